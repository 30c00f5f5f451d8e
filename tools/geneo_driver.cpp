// geneo_driver: the reference CLI driver's counterpart (src/geneo4PETSc.cpp main():1569) as a native executable over
// libgeneopc.so -- all of it lives in the library (csrc/driver_main.cpp, GeneoDriverMain); built by csrc/Makefile into
// geneo4petsc_amd/geneo_driver.
extern "C" int GeneoDriverMain(int argc, const char* const* argv);
int main(int argc, char** argv) { return GeneoDriverMain(argc - 1, argv + 1); }
