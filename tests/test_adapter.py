"""The PETSc-side adapter (adapters/geneo_petsc_adapter.cpp) defines what the reference's driver binds -- PETSc-typed
initGenEOPC (hdr/geneo.hpp:30-35, called at src/geneo4PETSc.cpp:1346), usageGenEO (:1566), createGenEOPC / PCGenEOSetup
(hdr/geneo_c.h:9-10) and a pc->data that IS the reference's geneoContext (read at :928-989 and :1123-1225).  PETSc is
installed neither here nor on the GPU box, so the adapter is COMPILED (type-checked, no link) against a declaration-only
stub of the PETSc / MPI names it uses (tests/petsc_stub) together with the REFERENCE's own hdr/geneo.hpp and
hdr/geneo_c.h, read where they lie (authoring container only: skipped where /root/reference is absent)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HDR = "/root/reference/hdr"
ADAPTER = os.path.join(ROOT, "adapters", "geneo_petsc_adapter.cpp")
STUB = os.path.join(ROOT, "tests", "petsc_stub")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_HDR, "geneo.hpp")), reason="reference headers not present")
def test_adapter_compiles_against_reference_headers_and_petsc_stub(tmp_path):
    obj = str(tmp_path / "adapter.o")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I", STUB, "-I", REF_HDR,
                           "-c", ADAPTER, "-o", obj])
    syms = subprocess.run(["nm", "-C", "--defined-only", obj], capture_output=True, text=True, check=True).stdout
    # the four names the reference driver links against, with the reference's (PETSc-typed) signatures
    assert " T createGenEOPC" in syms and " T PCGenEOSetup" in syms
    assert "initGenEOPC(_p_PC*&, unsigned int const&, unsigned int const&, _p_ISLocalToGlobalMapping* const&, _p_Mat* const&, " \
           "_p_Mat* const&, _p_Vec* const&, _p_Vec* const&" in syms
    assert "usageGenEO" in syms and "bool" in [l for l in syms.splitlines() if "usageGenEO" in l][0]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_HDR, "geneo.hpp")), reason="reference headers not present")
def test_reference_driver_accesses_compile_against_the_adapter_context(tmp_path):
    """Every public member of geneoContext the reference driver reads through pc->data (driver:928-989, :1123-1225)
    exists in what the adapter stores there -- it stores the reference's own class, so this is checked by compiling the
    same accesses."""
    src = tmp_path / "reads.cpp"
    src.write_text(r'''
#include <geneo.hpp>
double reads(PC pc) {
  geneoContext* g = (geneoContext*)pc->data;                 // src/geneo4PETSc.cpp:928, :1123
  PC l1; KSPGetPC(g->pcKSPL1Loc, &l1);                       // :946
  double t = g->optim + g->tau + g->gamma + g->lvl1SetupMinvTimeLoc + g->lvl2SetupTauLocTimeLoc + g->lvl2SetupTauSylTimeLoc +
             g->lvl2SetupTauEigTimeLoc + g->lvl2SetupGammaLocTimeLoc + g->lvl2SetupGammaSylTimeLoc + g->lvl2SetupGammaEigTimeLoc +
             g->lvl2SetupSylTimeLoc + g->lvl2SetupEigTimeLoc + g->lvl2SetupZTimeLoc + g->lvl2SetupETimeLoc + g->lvl1ApplyTimeLoc +
             g->lvl1ApplyScatterTimeLoc + g->lvl1ApplyMinvTimeLoc + g->lvl1ApplyGatherTimeLoc + g->lvl1ApplyPrjFSTimeLoc +
             g->lvl1ApplyPrjFSZtTimeLoc + g->lvl1ApplyPrjFSEinvTimeLoc + g->lvl1ApplyPrjFSZTimeLoc + g->lvl2ApplyTimeLoc +
             g->lvl2ApplyZtTimeLoc + g->lvl2ApplyEinvTimeLoc + g->lvl2ApplyZTimeLoc;
  return t + g->estimDimELoc + g->realDimELoc + g->nicolaidesLoc + g->lvl2 + (g->lvl1ORAS ? 1 : 0) + (g->effHybrid ? 1 : 0) +
         (g->hybrid ? 1 : 0) + (g->noSyl ? 1 : 0) + (g->offload ? 1 : 0) + g->name.size() + g->infoL2.size();
}
''')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", STUB, "-I", REF_HDR, "-fsyntax-only", str(src)])
