/* oracle/asan_main.c - driver of the sanitizer builds of the CPU oracle (`make -C oracle asan`; tests/test_oracle_sanitizers.py).  TEST INFRASTRUCTURE ONLY.
 * Rolls every model (U, R, P, Q, V, W) through resets and steps under the two action distributions of SURVEY.md 8d - B (workspace-uniform) and A (the literal
 * U(action_space) rollout, in which the hull scans, GJK, its cached simplices and the contact cache's first-contact insertions run all the time) - plus a state
 * round trip through the cache-row export / import and a threaded rpo_bench_rollout; AddressSanitizer / UBSan abort the process on the first finding.
 *     rpo_asan [steps] [envs]        prints one line per model, exits 0 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rp_oracle.h"

static unsigned long long s_rng = 0x9E3779B97F4A7C15ULL;
static double urand(void) { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return (double)(s_rng >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 40, envs = argc > 2 ? atoi(argv[2]) : 3;
  static const char* names[6] = {"U", "R", "P", "Q", "V", "W"};
  for (int kind = 0; kind < 6; kind++) {
    double worst = 0; long calls = 0;
    for (int dist = 0; dist < 2; dist++)
      for (int ei = 0; ei < envs; ei++) {
        rpo_env* e = rpo_create(kind, 1234, ei);
        rpo_obs o; double r, tp[7]; int ok;
        rpo_reset(e, 0, 0, &o);
        const int na = rpo_action_dim(e);
        double cfg[27]; rpo_get_config(e, cfg);
        float* row = (float*)malloc(sizeof(float) * (size_t)rpo_cache_row_words());
        for (int t = 0; t < steps; t++) {
          double a[10];
          for (int k = 0; k < na; k++) {
            const double hi = cfg[19 + k];
            if (dist == 1) a[k] = (2 * urand() - 1) * hi;                    /* distribution A: U(action_space) */
            else a[k] = k < 3 ? (k == 1 ? 0.3 * urand() : 0.36 * urand() - 0.18) : (k < na - 1 ? urand() - 0.5 : 2 * urand() - 1);
          }
          rpo_step(e, a, &o, &r, &ok, tp);
          if (t % 7 == 3) {                                                  /* the cache through its row form and back: nothing may change */
            rpo_get_cache_row(e, row);
            if (rpo_set_cache_row(e, row) != 0) { fprintf(stderr, "cache row rejected\n"); return 2; }
          }
          for (int k = 0; k < o.n_obs; k++) if (!isfinite(o.obs_quat[k])) { fprintf(stderr, "non-finite observation: kind %s env %d step %d\n", names[kind], ei, t); return 3; }
          if (fabs(o.obs_quat[0]) > worst) worst = fabs(o.obs_quat[0]);
        }
        long st[8]; rpo_gjk_stats(st, 1); calls += st[0];
        free(row);
        rpo_destroy(e);
      }
    printf("%s: %d envs x %d steps x 2 distributions clean, GJK calls %ld, max |ee x| %.3f\n", names[kind], envs, steps, calls, worst);
  }
  {                                                                          /* the threaded baseline leg of bench.py */
    const int n_envs = 8, n_steps = 6, na = 7;
    double* acts = (double*)malloc(sizeof(double) * n_envs * n_steps * na);
    for (int i = 0; i < n_envs * n_steps * na; i++) acts[i] = (2 * urand() - 1) * ((i % na) == na - 1 ? 1.0 : 6.0);
    const double t = rpo_bench_rollout(0, 99, n_envs, n_steps, na, acts, 4, -1.0);
    free(acts);
    if (!(t > 0)) { fprintf(stderr, "rpo_bench_rollout failed\n"); return 4; }
    printf("threads: 8 envs x 6 steps of distribution A on 4 threads clean\n");
  }
  return 0;
}
