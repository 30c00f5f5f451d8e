"""The CLI driver's INFO:/TIME: lines (reference shapes, driver:898-1231) -- host logic only (hostsim)."""
import io
import os
import re

import numpy as np

import dummy_cases as dc
import hostsim_util as hu
from geneo4petsc_amd import driver


def test_info_lines_match_the_dummy_goldens(tmp_path):
    """Lines 0-1 and the solve line are byte-identical to tst/dummy/*.ref; line 2 keeps the reference's
    token layout (tst/plot.py:74-95 parses it positionally) with this build's solver names."""
    d = dc.load()
    rec = [r for r in dc.geneo_refs() if r["file"] == "tridiag-pc=geneoASMH1-metis=nodal-opt=overlap1.ref"][0]
    inp = tmp_path / "tridiag.inp"
    inp.write_text(d["inputs"]["tridiag.inp"])
    ep, npart = dc.partition_for(rec)
    pf = tmp_path / "part.txt"
    pf.write_text("\n".join(str(v) for v in npart))
    buf = io.StringIO()
    lines, x = driver.run(["--inpFileA", str(inp), "--inpEps", "1.", "--np", "2", "--partFile", str(pf), "--metisNodal",
                           "--addOverlap", "1", "--shortRes", "-geneo_lvl", "ASM,H1", "-geneo_cut", "10",
                           "-ksp_rtol", "1e-12", "-ksp_atol", "1e-12"], lib=hu.hostsim_lib(), out=buf)
    assert lines[0] == rec["info"][0]
    assert lines[1] == rec["info"][1]
    assert lines[2].startswith("INFO: geneo1HASM pc, L1 ") and ", tau 0.10, L2 " in lines[2]
    assert lines[3] == rec["info"][3] == "INFO: solve - converged"
    np.testing.assert_allclose(x, rec["x"], rtol=1e-5)


def test_full_lines_parse_like_plot_py():
    buf = io.StringIO()
    lines, _ = driver.run(["--inpLibA", "laplacian#--size#8#--dim#3", "--np", "8", "--parts", "2,2,2", "--metisNodal",
                           "--addOverlap", "1", "--timing", "-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-geneo_cut", "4",
                           "-ksp_type", "cg"], lib=hu.hostsim_lib(), out=buf)
    info = [l for l in lines if l.startswith("INFO:")]
    # the token positions tst/plot.py relies on
    l0 = info[0].split()
    assert l0[l0.index("DOFs") + 1] == "512," and l0[l0.index("metis") + 1] == "nodal"
    l2 = info[2].split()
    assert l2[l2.index("pc,") - 1] == "geneo1ASM" and l2[l2.index("tau") + 1] == "0.20,"
    l3 = info[3].split()
    assert int(l3[l3.index("real") + 2]) == 32 or int(l3[l3.index("real") + 2]) > 0
    l4 = info[4].split()
    assert int(l4[5].replace(",", "")) > 0                      # iteration count position (plot.py:98)
    t = [l for l in lines if l.startswith("TIME:")][0].split()
    for pos in (3, 8, 12, 17, 21, 24):                          # plot.py:100-105
        float(t[pos].replace(",", ""))
    assert re.match(r"^      L1       setup: Minv \d+\.\d{5} s$", [l for l in lines if "L1       setup" in l][0])


def test_inp_lib_plugin_abi(tmp_path):
    """--inpLibA with a real getInput plugin (driver:75-96): the reference's own generator .so files, compiled
    from /root/reference into oracle/_ref (test infrastructure), load unchanged through GeneoGetLibInput."""
    import pytest
    from oracle import ref_generators as rg
    from geneo4petsc_amd import decomp
    if not rg.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = hu.hostsim_lib()
    for name, args, kw in (("laplacian", "--size#6#--dim#3#--kappa#10#lin", dict(size=6, dim=3, kappa_max=10.0, interp="lin")),
                           ("heat", "--size#5#--dim#2", dict(size=5, dim=2, heat=True)),
                           ("graph", "--size#30#--level#2#--noGround", None)):
        path = os.path.join(rg.REF_DIR, "lib%s.so" % name)
        m = decomp.plugin_mesh(lib, path, args)
        ours = decomp.graph_mesh(size=30, level=2, no_ground=True) if kw is None else decomp.grid_mesh(**kw)
        assert m.nbNode == ours.nbNode and np.array_equal(m.nodes, ours.nodes)
        np.testing.assert_allclose(m.mats, ours.mats, rtol=1e-15, atol=0)
    with pytest.raises(RuntimeError, match="open library KO|get input"):
        decomp.plugin_mesh(lib, str(tmp_path / "missing.so"), "")
    path = os.path.join(rg.REF_DIR, "liblaplacian.so")
    lines, x = driver.run(["--inpLibA", path + "#--size#8#--dim#3", "--np", "8", "--metisNodal", "--addOverlap", "1",
                           "-geneo_lvl", "ASM,1", "-geneo_tau", "0.2", "-geneo_cut", "4", "-ksp_type", "cg"],
                          lib=lib, out=io.StringIO())
    assert lines[0].startswith("INFO: nb DOFs 512, nb elements 1408,") and "converged" in lines[-1]


def test_inp_file_b_and_verbose(tmp_path):
    """--inpFileB (idx [value] lines, driver:841-858) and --verbose 1 ("The solution X is:" block of the goldens)."""
    d = dc.load()
    rec = [r for r in dc.geneo_refs() if r["use_b_file"] and r["geneo_lvl"].startswith("ASM")][0]      # identity.inp + B.inp
    inp = tmp_path / "A.inp"
    inp.write_text(d["inputs"][rec["input"] + ".inp"])
    bf = tmp_path / "B.inp"
    bf.write_text(d["inputs"]["B.inp"])
    ep, npart = dc.partition_for(rec)
    pf = tmp_path / "part.txt"
    part = ep if rec["metis"] == "dual" else npart
    pf.write_text("\n".join(str(v) for v in part))
    argv = ["--inpFileA", str(inp), "--inpFileB", str(bf), "--inpEps", str(rec["inpEps"]), "--np", "2", "--partFile", str(pf),
            "--metisDual" if rec["metis"] == "dual" else "--metisNodal", "--addOverlap", str(rec["overlap"]), "--verbose", "1",
            "-geneo_lvl", rec["geneo_lvl"], "-ksp_rtol", "1e-12", "-ksp_atol", "1e-12"]
    if rec["geneo_cut"] > 0:
        argv += ["-geneo_cut", str(rec["geneo_cut"])]
    lines, x = driver.run(argv, lib=hu.hostsim_lib(), out=io.StringIO())
    i = lines.index("The solution X is:")
    printed = np.array([float(v) for v in lines[i + 1:i + 1 + len(rec["x"])]])
    np.testing.assert_allclose(printed, rec["x"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x, rec["x"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# The native driver (csrc/driver_main.cpp, GeneoDriverMain; executable geneo4petsc_amd/geneo_driver): the reference CLI's
# counterpart in C++ over the C ABI (north_star: "host code stays C++").  Its INFO: lines must be those of the Python
# driver above, character by character; TIME: lines have the same shapes.
def _native(lib, argv, capfd):
    import ctypes as C
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    capfd.readouterr()
    rc = lib.GeneoDriverMain(len(argv), arr)
    out = capfd.readouterr()
    return rc, out.out.splitlines(), out.err


def test_native_driver_prints_the_python_drivers_lines(capfd):
    lib = hu.hostsim_lib()
    argv = ["--inpLibA", "laplacian#--size#8#--dim#3", "--np", "8", "--parts", "2,2,2", "--metisNodal", "--addOverlap", "1",
            "--timing", "--cmdLine", "-geneo_lvl", "SRAS,E1", "-geneo_tau", "0.2", "-geneo_cut", "4", "-ksp_type", "gmres"]
    rc, got, err = _native(lib, argv, capfd)
    assert rc == 0, err
    want, _ = driver.run(argv, lib=lib, out=io.StringIO())
    assert [l for l in got if l.startswith(("INFO:", "CMD:"))] == [l for l in want if l.startswith(("INFO:", "CMD:"))]
    shape = lambda l: re.sub(r"\d+\.\d+", "#", l)
    assert [shape(l) for l in got if not l.startswith(("INFO:", "CMD:"))] == [shape(l) for l in want if not l.startswith(("INFO:", "CMD:"))]
    # the built-in k-way partitioner (no --parts): same lines again, dual graph
    argv = ["--inpLibA", "heat#--size#7#--dim#2#--kappa#10#minmax", "--np", "3", "--metisDual", "--addOverlap", "2",
            "-geneo_lvl", "SORAS,2", "-geneo_tau", "0.05", "-geneo_gamma", "1.5", "-geneo_optim", "0.5", "-ksp_type", "cg"]
    rc, got, err = _native(lib, argv, capfd)
    assert rc == 0, err
    want, _ = driver.run(argv, lib=lib, out=io.StringIO())
    assert got == want


def test_native_driver_reads_the_reference_text_inputs(tmp_path, capfd):
    """--inpFileA / --inpFileB / --partFile / --verbose through the C++ readers (driver:98-194, :841-858): the tst/dummy
    golden's INFO lines byte for byte and its converged solution."""
    lib = hu.hostsim_lib()
    d = dc.load()
    rec = [r for r in dc.geneo_refs() if r["file"] == "tridiag-pc=geneoASMH1-metis=nodal-opt=overlap1.ref"][0]
    inp = tmp_path / "tridiag.inp"
    inp.write_text(d["inputs"]["tridiag.inp"])
    ep, npart = dc.partition_for(rec)
    pf = tmp_path / "part.txt"
    pf.write_text("\n".join(str(v) for v in npart))
    rc, got, err = _native(lib, ["--inpFileA", str(inp), "--inpEps", "1.", "--np", "2", "--partFile", str(pf), "--metisNodal",
                                 "--addOverlap", "1", "--shortRes", "--verbose", "1", "-geneo_lvl", "ASM,H1", "-geneo_cut", "10",
                                 "-ksp_rtol", "1e-12", "-ksp_atol", "1e-12"], capfd)
    assert rc == 0, err
    info = [l for l in got if l.startswith("INFO:")]
    assert info[0] == rec["info"][0] and info[1] == rec["info"][1] and info[3] == rec["info"][3] == "INFO: solve - converged"
    i = got.index("The solution X is:")
    np.testing.assert_allclose([float(v) for v in got[i + 1:i + 1 + len(rec["x"])]], rec["x"], rtol=1e-5)
    # identity.inp + B.inp
    rec = [r for r in dc.geneo_refs() if r["use_b_file"] and r["geneo_lvl"].startswith("ASM")][0]
    inp = tmp_path / "A.inp"
    inp.write_text(d["inputs"][rec["input"] + ".inp"])
    bf = tmp_path / "B.inp"
    bf.write_text(d["inputs"]["B.inp"])
    ep, npart = dc.partition_for(rec)
    pf.write_text("\n".join(str(v) for v in (ep if rec["metis"] == "dual" else npart)))
    argv = ["--inpFileA", str(inp), "--inpFileB", str(bf), "--inpEps", str(rec["inpEps"]), "--np", "2", "--partFile", str(pf),
            "--metisDual" if rec["metis"] == "dual" else "--metisNodal", "--addOverlap", str(rec["overlap"]), "--verbose", "1",
            "-geneo_lvl", rec["geneo_lvl"], "-ksp_rtol", "1e-12", "-ksp_atol", "1e-12"]
    if rec["geneo_cut"] > 0:
        argv += ["-geneo_cut", str(rec["geneo_cut"])]
    rc, got, err = _native(lib, argv, capfd)
    assert rc == 0, err
    i = got.index("The solution X is:")
    np.testing.assert_allclose([float(v) for v in got[i + 1:i + 1 + len(rec["x"])]], rec["x"], rtol=1e-5, atol=1e-6)
    # errors come back as a return code and a message, as the reference's `Error: ...` lines
    rc, got, err = _native(lib, ["--inpFileA", str(tmp_path / "missing.inp"), "--np", "2"], capfd)
    assert rc == 1 and err.startswith("Error:")


def test_native_driver_loads_a_reference_getinput_plugin(capfd):
    from oracle import ref_generators as rg
    import pytest
    if not rg.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = hu.hostsim_lib()
    argv = ["--inpLibA", os.path.join(rg.REF_DIR, "libgraph.so") + "#--size#30#--level#2#--noGround", "--np", "4", "--metisNodal",
            "--addOverlap", "1", "-geneo_lvl", "ASM,1", "-geneo_tau", "0.3", "-geneo_cut", "6", "-ksp_type", "cg"]
    rc, got, err = _native(lib, argv, capfd)
    assert rc == 0, err
    want, _ = driver.run(argv, lib=lib, out=io.StringIO())
    assert got == want


import pytest   # noqa: E402


@pytest.mark.gpu
def test_native_driver_executable_on_the_gpu():
    """geneo4petsc_amd/geneo_driver (a main() over GeneoDriverMain, linked against libgeneopc.so) on the MI355X."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "geneo4petsc_amd", "geneo_driver")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe, "--inpLibA", "laplacian#--size#24#--dim#3", "--np", "8", "--parts", "2,2,2", "--metisNodal", "--addOverlap",
                        "2", "--timing", "-geneo_lvl", "SRAS,1", "-geneo_tau", "0.35", "-geneo_cut", "20", "-ksp_type", "cg"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert lines[0].startswith("INFO: nb DOFs 13824,") and "INFO: geneo1SRAS pc, L1 pcg-amg no-proj-fine-space, tau 0.35, L2 lobpcg cholesky" in lines
    solve = [l for l in lines if l.startswith("INFO: solve - ")][0]
    assert "converged (KSP_CONVERGED_RTOL), 25 iteration(s)" in solve       # tests/test_bench_options.py: 24^3 at these options
    assert [l for l in lines if l.startswith("TIME: read input")]
