/*
 * clipper_abi.h — plain-C types shared by the HIP product library (clipper_hip.h)
 * and by the CPU oracle (oracle/clipper_ref.h), so that tests can drive both
 * through identical argument lists.
 *
 * Mirrors (reference, /root/reference):
 *   struct clipper::Params            include/clipper/clipper.h:27-60
 *   enum   clipper::Params::Rounding  include/clipper/clipper.h:49-59
 *   struct clipper::Solution          include/clipper/clipper.h:65-73
 *
 * Everything crossing this boundary is a POD, a plain pointer or a size.
 * Matrices are column-major fp64 (Eigen's default layout, types.h:15-23);
 * the association list A is column-major m x 2 int32 (column 0 = index into
 * D1 for all m rows, then column 1 = index into D2), exactly the memory
 * layout of Eigen::Matrix<int, Dynamic, 2>.
 */
#ifndef CLIPPER_ABI_H
#define CLIPPER_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* clipper::Params::Rounding (clipper.h:49-59) */
enum {
  CLIPPER_ROUNDING_NONZERO = 0,
  CLIPPER_ROUNDING_DSD     = 1, /* exact densest sub-graph of nnz(u): Goldberg, on the host    */
  CLIPPER_ROUNDING_DSD_HEU = 2
};

/* clipper::Params with the reference's defaults documented (clipper.h:27-60). */
typedef struct clipper_params_t {
  double tol_u;       /* 1e-8   stop when ||unew-u|| < tol_u                 */
  double tol_F;       /* 1e-9   stop when |Fnew-F|   < tol_F                 */
  double tol_Fop;     /* 1e-10  declared by the reference, never read        */
  int32_t maxiniters; /* 200    gradient-ascent steps per penalty value      */
  int32_t maxoliters; /* 1000   outer (penalty homotopy) iterations          */
  double beta;        /* 0.25   backtracking factor                          */
  int32_t maxlsiters; /* 99     line-search trials per step                  */
  double eps;         /* 1e-9   numerical zero                               */
  double affinityeps; /* 1e-4   affinity sparsification threshold            */
  int32_t rescale_u0; /* 1      one power-method step on u0                  */
  int32_t rounding;   /* CLIPPER_ROUNDING_DSD_HEU                            */
} clipper_params_t;

/* Scalar part of clipper::Solution plus the pass counter this build reports. */
typedef struct clipper_solve_info_t {
  double  score;     /* final objective F (Solution::score)                         */
  double  seconds;   /* wall time of the solve (Solution::t)                        */
  double  d;         /* final penalty value (diagnostic; not in the reference)      */
  int32_t ifinal;    /* outer iterations executed (Solution::ifinal)                */
  int32_t num_nodes; /* |Solution::nodes|                                           */
  int64_t n_passes;  /* passes over M actually executed (one pass yields both
                        M_off*x and C_off*x)                                        */
  int64_t n_trials;  /* line-search trials (gradient evaluations inside the loops)  */
} clipper_solve_info_t;

/* Fills `p` with the reference's defaults (clipper.h:27-60). Header-only so that
 * both libraries and C callers agree without a link dependency. */
static inline void clipper_params_default(clipper_params_t* p) {
  p->tol_u = 1e-8;
  p->tol_F = 1e-9;
  p->tol_Fop = 1e-10;
  p->maxiniters = 200;
  p->maxoliters = 1000;
  p->beta = 0.25;
  p->maxlsiters = 99;
  p->eps = 1e-9;
  p->affinityeps = 1e-4;
  p->rescale_u0 = 1;
  p->rounding = CLIPPER_ROUNDING_DSD_HEU;
}

#ifdef __cplusplus
}
#endif
#endif /* CLIPPER_ABI_H */
