/* selftest.c -- sanitizer driver for the CPU restatement (TEST INFRASTRUCTURE).
 * Built by tests/test_oracle_sanitizers.py together with tb_oracle.c under
 * -fsanitize=address,undefined and run as a plain process:
 *   selftest <params.bin> <env_kind> <n_envs> <steps>
 * params.bin holds the raw bytes of one TbParams. Random actions from a fixed LCG; with
 * auto-reset on, so resets, contacts, the fast-forward and the solver all run instrumented. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tb_oracle.h"

int main(int argc, char **argv) {
  if (argc != 5) { fprintf(stderr, "usage: selftest params.bin kind n steps\n"); return 2; }
  TbParams P;
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(&P, sizeof P, 1, f) != 1) { fprintf(stderr, "cannot read TbParams\n"); return 2; }
  fclose(f);
  int kind = atoi(argv[2]), n = atoi(argv[3]), steps = atoi(argv[4]);
  int od = kind == TB_ENV_SWING ? TB_SWING_OBS_DIM : TB_TENNIS_OBS_DIM, ad = kind == TB_ENV_SWING ? TB_SWING_ACT_DIM : TB_TENNIS_ACT_DIM;
  P.flags |= TB_F_AUTO_RESET;
  TboBatch *b = tbo_create(&P, kind, n, 12345u, 0);
  if (!b) return 3;
  float *obs = malloc(sizeof(float) * n * od), *act = malloc(sizeof(float) * n * ad), *rew = malloc(sizeof(float) * n), *term = malloc(sizeof(float) * n * od);
  uint8_t *done = malloc(n);
  int32_t *sub = malloc(sizeof(int32_t) * n);
  uint32_t *words = malloc(sizeof(uint32_t) * TB_SWING_WORDS * n);
  tbo_reset(b, NULL, obs);
  uint32_t lcg = 1u;
  double acc = 0;
  for (int t = 0; t < steps; ++t) {
    for (int i = 0; i < n * ad; ++i) { lcg = lcg * 1664525u + 1013904223u; act[i] = (float)(lcg >> 8) / 8388608.0f - 1.0f; }
    tbo_step(b, act, obs, rew, done, term, sub);
    for (int i = 0; i < n; ++i) acc += rew[i];
  }
  tbo_get_state(b, words, done);
  tbo_set_state(b, words, done);
  uint64_t c[TB_N_COUNTERS];
  tbo_counters(b, c);
  printf("reward sum %.3f substeps %llu episodes %llu nonfinite %llu\n", acc, (unsigned long long)c[6], (unsigned long long)c[5], (unsigned long long)c[7]);
  tbo_destroy(b);
  free(obs); free(act); free(rew); free(term); free(done); free(sub); free(words);
  return c[7] ? 4 : 0;
}
