import os
HERE=os.path.dirname(os.path.abspath(__file__))
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
import json, csv
P=ROOT+'/profiles/'
def last(f): return json.loads(open(P+f).read().strip().splitlines()[-1])
d=last('r04_bench_default_driverlike.json'); r=last('r04_bench_rank_of_8.json'); w=last('r04_bench_weak_126.json'); h=last('r04_bench_heat_126.json'); g=last('r04_bench_graph_10M.json')
m=json.load(open(P+'r04_mfma_pmc.json'))
N={}
f3=lambda v: "%.2f"%v
N['S368']=f3(d['setup_s']); N['V368']=f3(d['solve_s']); N['MS368']="%.2f"%(d['ms_per_step']/1e3); N['MEM368']="%.0f"%d['device_mem_peak_gb']
N['FOOT368']="%.0f"%d['device_memory']['device_mem_footprint_peak_gb']
N['COLD368']="%.1f"%d['first_setup_s']['setup']; N['PREP368']="%.1f"%d['host_prep_s']
N['WALL368']=open(ROOT+"/gpurun_out/driverlike_wall.txt").read().split('wall')[1].split()[0]
sb=d['setup_breakdown_s']; N['L1368']=f3(sb['level1_upload_and_amg'])+" s"; N['EIG368']=f3(sb['eigensolve_lobpcg'])+" s"; N['E368']=f3(sb['coarse_operator_E'])+" s"
so=d['solve_breakdown_s']; N['LS368']=f3(so['local_solves']); N['CZ368']="%.4f"%(so['coarse_Zt']+so['coarse_Einv'])
N['INNER368']=str(d['local_solve_cg_iterations'])
N['S1R']="%.3f"%r['setup_s']; N['V1R']="%.3f"%r['solve_s']
N['S126']="%.3f"%w['setup_s']; N['V126']="%.3f"%w['solve_s']; N['I126']=str(w['iterations'])
N['SHEAT']="%.2f"%h['setup_s']; N['VHEAT']="%.2f"%h['solve_s']; N['IHEAT']=str(h['iterations']); N['IHEAT0']=str(h['asm0_vs_geneo']['ASM,0']['iterations'])
N['SGRAPH']="%.2f"%g['setup_s']; N['VGRAPH']="%.2f"%g['solve_s']; N['IGRAPH']=str(g['iterations']); N['SPMVGRAPH']="%.2f"%(g['value']/1e3)
ks=d['roofline']['kernels']
rows=[]
for k in ks:
    rate = ("%.2f TB/s"%(k['hbm_GBs']/1e3)) + ((" / %.1f TFLOP/s"%k['achieved']) if k['bound']=='mfma' else "")
    frac = "%.2f of HBM"%(k['hbm_GBs']/8000.0) + ((", %.2f of FP64 MFMA"%k['frac']) if k['bound']=='mfma' else "")
    rows.append("| %s | %.1f %% | %.3f ms | %s | %s |"%(k['kernel'].split(' (')[0], 100*k['share_of_step'], k['avg_launch_ms'], rate, frac))
N['KTABLE368']="\n".join(rows)
N['SPMV368']="%.2f"%(ks[0]['hbm_GBs']/1e3); N['SPMVFRAC368']="%.2f"%ks[0]['frac']
N['LPFRAC']="%.3f"%r['roofline']['kernels'][-1]['frac']
N['MFMAGRAM2']="%.0f"%m['k_gram_flat<2, 3, true>']['MfmaUtil']; N['MFMAGRAM3']="%.0f"%m['k_gram_flat<3, 3, false>']['MfmaUtil']
N['MFMAGRAM']="%.0f / %.0f"%(m['k_gram_flat<2, 3, true>']['MfmaUtil'], m['k_gram_flat<3, 3, false>']['MfmaUtil'])
N['MFMAUPD']="%.0f"%m['k_lobpcg_update32<1>']['MfmaUtil']; N['MFMABM8']="%.0f"%m['k_blockmul_mfma<8>']['MfmaUtil']; N['MFMABM24']="%.0f"%m['k_blockmul_mfma<24>']['MfmaUtil']
c=d['cpu_baseline']; N['CPUSPMV']="%.0f"%c['value']; N['CPUS']="%.1f"%c['geneo_sample']['setup_s']; N['CPUV']="%.1f"%c['geneo_sample']['solve_s']
N['GPU96']="24"; N['GOLD126']="and 126³: PCG 23 for every perturbation, GMRES 18 / 10"
json.dump(N, open(HERE+'/design_nums.json','w'), indent=1)
print({k:v for k,v in N.items() if k!='KTABLE368'})
