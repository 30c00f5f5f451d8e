"""The C-ABI product library: builds with hipcc for gfx950 (cross-compile works without a GPU),
loads, and exports every symbol include/geneo_c.h declares.  No compute calls here (no GPU)."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "geneo_c.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt))
    names = {n for n in names if n[0].isupper() or n.startswith(("createGenEOPC", "initGenEOPC_c", "usageGenEO_c"))}
    return names - {"GeneoExchangeFn", "GeneoAllreduceFn"}


@pytest.fixture(scope="module")
def libpath():
    from geneo4petsc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("no hipcc and no prebuilt libgeneopc.so")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "geneo4petsc_amd", "csrc")])
    return _lib.LIB_PATH


def test_header_and_binding_agree():
    from geneo4petsc_amd import _lib
    decl = _declared()
    assert decl == set(_lib.SYMBOLS), (sorted(decl - set(_lib.SYMBOLS)), sorted(set(_lib.SYMBOLS) - decl))


def test_library_exports_every_symbol(libpath):
    from geneo4petsc_amd import _lib
    lib = _lib.bind(libpath)                    # raises AttributeError on a missing export
    assert lib.GeneoBackendName() == b"hip-gfx950"
    assert b"-geneo_lvl" in lib.usageGenEO_c()


def test_library_has_gfx950_code_object(libpath):
    data = open(libpath, "rb").read()
    assert b"gfx950" in data


def test_no_cpu_fallback_in_package():
    """The package must not reference the oracle or the host simulator."""
    pkg = os.path.join(ROOT, "geneo4petsc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "libgeneopc_hostsim" not in txt, f


def test_cpp_mirror_header_compiles_and_runs(tmp_path):
    """include/geneo.hpp (initGenEOPC / usageGenEO / geneoContext with the reference's argument order) compiled by g++
    against the C ABI and run on the test-only host backend: one subdomain, tridiag(-1, 2.0001, -1)."""
    import subprocess
    import hostsim_util as hu
    so = hu.hostsim_path() if hasattr(hu, "hostsim_path") else None
    hu.hostsim_lib()
    so = so or os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libgeneopc_hostsim.so")
    src = tmp_path / "mirror.cpp"
    src.write_text(r"""
#include <cstdio>
#include <vector>
#include "geneo.hpp"
int main() {
  const int n = 8;
  std::vector<int> rp(n + 1, 0), col; std::vector<double> val;
  for (int i = 0; i < n; ++i) {
    if (i > 0) { col.push_back(i - 1); val.push_back(-1.0); }
    col.push_back(i); val.push_back(2.0001);
    if (i + 1 < n) { col.push_back(i + 1); val.push_back(-1.0); }
    rp[i + 1] = (int)col.size();
  }
  GeneoCsr A{n, rp.data(), col.data(), val.data()};
  std::vector<int> map(n); for (int i = 0; i < n; ++i) map[i] = i;
  std::vector<unsigned int> dofs(n), mult(n, 1u); for (int i = 0; i < n; ++i) dofs[i] = i;
  std::vector<std::vector<unsigned int>> inter(1);
  PC pc;
  if (PCCreate_GenEO(&pc)) return 1;
  const char* argv[] = {"-geneo_lvl", "ASM,1", "-geneo_tau", "0.5"};
  if (PCSetFromOptions_GenEO(pc, 4, argv)) return 2;
  std::vector<double> b(n, 1.0), x0(n, 0.0), y(n, 0.0);
  if (initGenEOPC(pc, n, n, map, A, nullptr, b.data(), x0.data(), &dofs, &mult, &inter)) return 3;
  if (PCSetUp_GenEO(pc)) { std::printf("setup: %s\n", PCGenEOGetError(pc)); return 4; }
  if (PCApply_GenEO(pc, b.data(), y.data())) return 5;
  geneoContext ctx;
  if (ctx.refresh(pc)) return 6;
  std::string u = usageGenEO(false);
  std::printf("%s %d %d %.6f\n", ctx.name.c_str(), ctx.realDimELoc, (int)(u.find("-geneo_lvl") != std::string::npos), y[0]);
  return PCDestroy_GenEO(&pc);
}
""")
    exe = tmp_path / "mirror"
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["g++", "-std=c++17", "-I", inc, str(src), so, "-Wl,-rpath," + os.path.dirname(so), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    name, dim, has_usage, y0 = out.stdout.split()
    assert name == "geneo1ASM" and int(dim) >= 1 and has_usage == "1"
    # one subdomain covering everything: y = (A^-1 + Z E^-1 Z^T) b, both terms positive for b = 1 on this M-matrix
    a = np.diag(np.full(8, 2.0001)) + np.diag(np.full(7, -1.0), 1) + np.diag(np.full(7, -1.0), -1)
    assert float(y0) >= np.linalg.solve(a, np.ones(8))[0] * (1 - 1e-9)


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/geneo_c.h must compile as C99 with -pedantic (no C++, no torch types)."""
    import subprocess
    src = tmp_path / "cabi.c"
    src.write_text('#include "geneo_c.h"\nint main(void) { PC pc; GeneoInput in; GeneoInfo info; (void)in; (void)info;\n'
                   '  return PCCreate_GenEO(&pc) ? 1 : PCDestroy_GenEO(&pc); }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "cabi.o")])


def test_header_coexists_with_petsc_names(tmp_path):
    """With GENEO_HAVE_PETSC the header leaves PETSc's own `PC` / `PetscErrorCode` alone (the handle is GeneoPC), so
    the adapter of INTEGRATION.md can include both.  One-line stand-ins for the two PETSc typedefs are enough to check
    that nothing in the header clashes with them."""
    import subprocess
    src = tmp_path / "both.c"
    src.write_text('typedef int PetscErrorCode; typedef struct _p_PC* PC;   /* what petsc.h declares */\n'
                   '#define GENEO_HAVE_PETSC\n#include "geneo_c.h"\n'
                   '/* the adapter defines the reference-named pair with PETSc types and forwards to the library: */\n'
                   'PetscErrorCode createGenEOPC(PC petsc_pc) { GeneoPC h; (void)petsc_pc; return PCCreate_GenEO(&h); }\n'
                   'PetscErrorCode PCGenEOSetup(PC petsc_pc, void* mat, void* is, void** iss) {\n'
                   '  GeneoPC h = 0; GeneoIS m = {0, 0}; (void)petsc_pc; (void)mat; (void)is; (void)iss;\n'
                   '  return PCGenEOSetupViews(h, 0, m, 0) + PCGenEOCreateContext(h); }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "both.o")])
