// geneo.hpp -- C++ mirror of the reference's hdr/geneo.hpp on top of the C ABI (geneo_c.h).
//
//   reference hdr/geneo.hpp:30-35   initGenEOPC(PC&, nbDOF, nbDOFLoc, pcMap, pcA, pcADirLoc, pcB, pcX0,
//                                               dofIdxDomLoc, dofIdxMultLoc, intersectLoc)
//   reference hdr/geneo.hpp:41      std::string usageGenEO(bool petscPrintf = true)
//   reference hdr/geneo.hpp:46-138  class geneoContext (public data the driver reads directly,
//                                   src/geneo4PETSc.cpp:928-989,:1123-1225)
//
// Without PETSc the Mat / Vec / ISLocalToGlobalMapping arguments become plain views: the MATIS local
// matrix as GeneoCsr, the map as an index vector, vectors as device pointers.  Header-only.
#ifndef __geneo__
#define __geneo__

#include <cstdio>
#include <string>
#include <vector>

#include "geneo_c.h"

/*
 * initGenEOPC: initialize the GenEO PC (same argument order and meaning as the reference).
 *   - pcPC: PC created by createGenEOPC (PCCreate_GenEO).
 *   - nbDOF, nbDOFLoc, pcMap: local/global mapping (ascending global ids of the local DOFs).
 *   - pcA: local (Neumann) matrix of the MATIS operator.   - pcADirLoc: optional local Dirichlet matrix.
 *   - pcB_dev: right hand side (device, owned DOFs), may be NULL.
 *   - pcX0_dev: initial guess (device); written by PCSetUp_GenEO: Q b for the efficient hybrid, 0 otherwise.
 *     The caller MUST treat the initial guess as non zero (KSPSetInitialGuessNonzero, hdr/geneo.hpp:21).
 *   - dofIdxDomLoc: global ids of the local DOFs (same content as pcMap, kept for signature parity).
 *   - dofIdxMultLoc: multiplicities, in the order of pcMap.
 *   - intersectLoc: per rank list of shared local DOFs (only used by GenEO-2).
 */
inline PetscErrorCode initGenEOPC(PC& pcPC, unsigned int const& nbDOF, unsigned int const& nbDOFLoc,
                                  std::vector<int> const& pcMap, GeneoCsr const& pcA, GeneoCsr const* pcADirLoc,
                                  double const* pcB_dev, double* pcX0_dev,
                                  std::vector<unsigned int> const* const dofIdxDomLoc,
                                  std::vector<unsigned int> const* const dofIdxMultLoc,
                                  std::vector<std::vector<unsigned int>> const* const intersectLoc) {
  (void)dofIdxDomLoc;
  if (!dofIdxMultLoc || dofIdxMultLoc->size() != nbDOFLoc || pcMap.size() != nbDOFLoc) return 1;
  PetscErrorCode rc = initGenEOPC_c(pcPC, nbDOF, nbDOFLoc, pcMap.data(), &pcA, pcADirLoc, pcB_dev, pcX0_dev,
                                    dofIdxMultLoc->data());
  if (rc || !intersectLoc) return rc;
  std::vector<int> nonempty(intersectLoc->size());     // GenEO-2 reads only the emptiness of each list (geneo.cpp:1139-1148)
  for (size_t q = 0; q < intersectLoc->size(); ++q) nonempty[q] = (*intersectLoc)[q].empty() ? 0 : 1;
  return PCGenEOSetIntersect(pcPC, -1, (int)nonempty.size(), nonempty.data());   // -1: the subdomain just added
}

/* usageGenEO: usage of GenEO (printf stands for PetscPrintf). */
inline std::string usageGenEO(bool const print = true) {
  std::string msg = usageGenEO_c();
  if (print) std::fputs(msg.c_str(), stdout);
  return msg;
}

/* The public counters / timers of the reference's geneoContext, refreshed from the library. */
class geneoContext {
 public:
  std::string name;
  int estimDimELoc = 0, realDimELoc = 0, nicolaidesLoc = 0;
  double lvl1SetupMinvTimeLoc = 0, lvl2SetupEigTimeLoc = 0, lvl2SetupZTimeLoc = 0, lvl2SetupETimeLoc = 0;
  double lvl1ApplyTimeLoc = 0, lvl1ApplyScatterTimeLoc = 0, lvl1ApplyMinvTimeLoc = 0, lvl1ApplyGatherTimeLoc = 0;
  double lvl1ApplyPrjFSTimeLoc = 0, lvl2ApplyTimeLoc = 0, lvl2ApplyZtTimeLoc = 0, lvl2ApplyEinvTimeLoc = 0,
         lvl2ApplyZTimeLoc = 0;
  PetscErrorCode refresh(PC pc) {
    GeneoInfo i;
    PetscErrorCode rc = PCGenEOGetInfo(pc, &i);
    if (rc) return rc;
    name = PCGenEOGetName(pc);
    estimDimELoc = i.estimDimELoc; realDimELoc = i.realDimELoc; nicolaidesLoc = i.nicolaidesLoc;
    lvl1SetupMinvTimeLoc = i.lvl1SetupMinvTimeLoc; lvl2SetupEigTimeLoc = i.lvl2SetupEigTimeLoc;
    lvl2SetupZTimeLoc = i.lvl2SetupZTimeLoc; lvl2SetupETimeLoc = i.lvl2SetupETimeLoc;
    lvl1ApplyTimeLoc = i.lvl1ApplyTimeLoc; lvl1ApplyScatterTimeLoc = i.lvl1ApplyScatterTimeLoc;
    lvl1ApplyMinvTimeLoc = i.lvl1ApplyMinvTimeLoc; lvl1ApplyGatherTimeLoc = i.lvl1ApplyGatherTimeLoc;
    lvl1ApplyPrjFSTimeLoc = i.lvl1ApplyPrjFSTimeLoc; lvl2ApplyTimeLoc = i.lvl2ApplyTimeLoc;
    lvl2ApplyZtTimeLoc = i.lvl2ApplyZtTimeLoc; lvl2ApplyEinvTimeLoc = i.lvl2ApplyEinvTimeLoc;
    lvl2ApplyZTimeLoc = i.lvl2ApplyZTimeLoc;
    return 0;
  }
};

#endif
