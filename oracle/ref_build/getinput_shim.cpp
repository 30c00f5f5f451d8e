// TEST INFRASTRUCTURE ONLY -- ours, not reference code.  Calls a reference `getInput` plugin exactly as
// the reference driver does (src/geneo4PETSc.cpp:75-96: dlopen, dlsym "getInput", '#' -> ' ' in the
// argument string) and flattens the C++ containers into plain arrays for ctypes.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef int (*get_input_fn)(std::string const& args, unsigned int& nbElem, unsigned int& nbNode,
                            std::vector<unsigned int>& elemPtr, std::vector<unsigned int>& elemIdx,
                            std::vector<std::vector<double>>& elemSubMat);

extern "C" int ref_get_input(const char* lib_path, const char* args_c, unsigned int* nb_elem, unsigned int* nb_node,
                             unsigned int** elem_ptr, unsigned int* n_ptr, unsigned int** elem_idx, unsigned int* n_idx,
                             double** mats, unsigned int* n_mats) {
  void* lib = dlopen(lib_path, RTLD_LAZY | RTLD_LOCAL);
  if (!lib) return 1;
  get_input_fn fn = (get_input_fn)dlsym(lib, "getInput");
  if (!fn) { dlclose(lib); return 2; }
  std::string args = args_c;
  for (auto& ch : args)
    if (ch == '#') ch = ' ';
  unsigned int ne = 0, nn = 0;
  std::vector<unsigned int> ptr, idx;
  std::vector<std::vector<double>> sub;
  const int rc = fn(args, ne, nn, ptr, idx, sub);
  if (rc != 0) { dlclose(lib); return 3; }
  *nb_elem = ne;
  *nb_node = nn;
  *n_ptr = (unsigned int)ptr.size();
  *n_idx = (unsigned int)idx.size();
  *elem_ptr = (unsigned int*)malloc(sizeof(unsigned int) * (ptr.size() + 1));
  *elem_idx = (unsigned int*)malloc(sizeof(unsigned int) * (idx.size() + 1));
  memcpy(*elem_ptr, ptr.data(), sizeof(unsigned int) * ptr.size());
  memcpy(*elem_idx, idx.data(), sizeof(unsigned int) * idx.size());
  size_t tot = 0;
  for (auto& m : sub) tot += m.size();
  *n_mats = (unsigned int)tot;
  *mats = (double*)malloc(sizeof(double) * (tot + 1));
  size_t pos = 0;
  for (auto& m : sub) {
    memcpy(*mats + pos, m.data(), sizeof(double) * m.size());
    pos += m.size();
  }
  dlclose(lib);
  return 0;
}
extern "C" void ref_free(void* p) { free(p); }
