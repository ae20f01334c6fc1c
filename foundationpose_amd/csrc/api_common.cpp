// Error plumbing, mesh handles and the host-side cluster_poses of libfp_amd.so.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "fp_common.h"

static thread_local char g_err[512] = "";

void fp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fp_last_error(void) { return g_err; }
extern "C" int fp_version(void) { return FP_AMD_ABI_VERSION; }

extern "C" int fp_mesh_create(const float* pos, const float* nrm, const int32_t* faces, const float* uv,
                              const int32_t* uv_idx, const float* tex, const float* vcol, int V, int T, int Ht,
                              int Wt, fp_mesh** out) {
  FP_REQUIRE(out != nullptr, "fp_mesh_create: out is NULL");
  FP_REQUIRE(pos && nrm && faces, "fp_mesh_create: pos/nrm/faces are required");
  FP_REQUIRE(V > 0 && T > 0, "fp_mesh_create: empty mesh (V=%d, T=%d)", V, T);
  FP_REQUIRE((tex && uv && Ht > 0 && Wt > 0) || vcol, "fp_mesh_create: need (tex, uv) or vcol");
  fp_mesh* m = (fp_mesh*)calloc(1, sizeof(fp_mesh));
  m->pos = pos; m->nrm = nrm; m->faces = faces;
  m->uv = tex ? uv : nullptr;
  m->uv_idx = tex ? uv_idx : nullptr;
  m->tex = tex;
  m->vcol = tex ? nullptr : vcol;
  m->V = V; m->T = T; m->Ht = tex ? Ht : 0; m->Wt = tex ? Wt : 0;
  *out = m;
  return FP_OK;
}

extern "C" void fp_mesh_destroy(fp_mesh* mesh) { free(mesh); }

// Greedy symmetry-aware pose clustering (reference: mycpp/src/app/pybind_api.cpp:24-68,
// geodesic distance mycpp/src/Utils.cpp:21-26).  Init-time, O(N^2 S), host only.
extern "C" int fp_cluster_poses(float angle_diff_deg, float dist_diff, const float* poses, int N,
                                const float* sym, int S, int* keep_idx) {
  if (N <= 0) return 0;
  FP_REQUIRE(poses && sym && keep_idx && S > 0, "fp_cluster_poses: bad arguments");
  const float thres = (float)(angle_diff_deg / 180.0 * M_PI);
  int kept = 0;
  keep_idx[kept++] = 0;
  for (int i = 1; i < N; ++i) {
    const float* P = poses + (size_t)i * 16;
    bool fresh = true;
    for (int k = 0; k < kept && fresh; ++k) {
      const float* Q = poses + (size_t)keep_idx[k] * 16;
      const float ex = Q[3] - P[3], ey = Q[7] - P[7], ez = Q[11] - P[11];
      if (sqrtf((ex * ex + ey * ey) + ez * ez) >= dist_diff) continue;
      for (int s = 0; s < S; ++s) {
        const float* G = sym + (size_t)s * 16;
        float tr = 0.f;  // trace((P*G)_rot * Q_rot^T)
        for (int r = 0; r < 3; ++r) {
          float row[3];
          for (int c = 0; c < 3; ++c)
            row[c] = ((P[r * 4] * G[c] + P[r * 4 + 1] * G[4 + c]) + P[r * 4 + 2] * G[8 + c]) + P[r * 4 + 3] * G[12 + c];
          tr += (row[0] * Q[r * 4] + row[1] * Q[r * 4 + 1]) + row[2] * Q[r * 4 + 2];
        }
        float cs = fmaxf(fminf((tr - 1.0f) / 2.0f, 1.0f), -1.0f);
        if (acosf(cs) < thres) { fresh = false; break; }
      }
    }
    if (fresh) keep_idx[kept++] = i;
  }
  return kept;
}
