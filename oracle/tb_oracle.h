/*
 * tb_oracle.h -- API of the CPU restatement (TEST INFRASTRUCTURE; see tb_oracle.c).
 * Mirrors the product C ABI (include/tb_stepper.h) on host buffers so a parity test
 * reads: same TbParams, same actions in, compare obs / reward / done / state out.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef TB_ORACLE_H
#define TB_ORACLE_H

#include "../include/tb_stepper.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct TboBatch TboBatch;

TboBatch *tbo_create(const TbParams *params, int env_kind, int n_envs, uint64_t seed, uint64_t env_id_base);
void tbo_destroy(TboBatch *b);
void tbo_set_params(TboBatch *b, const TbParams *params);
void tbo_set_threads(TboBatch *b, int threads); /* OpenMP threads over envs; 1 = scalar */
int tbo_real_bytes(void);                       /* 4 (f32 build) or 8 (f64 build) */

void tbo_reset(TboBatch *b, const uint8_t *mask, float *obs);
void tbo_step(TboBatch *b, const float *actions, float *obs, float *reward, uint8_t *done,
              float *terminal_obs, int32_t *substeps);
void tbo_counters(TboBatch *b, uint64_t *out);
void tbo_counters_reset(TboBatch *b);

void tbo_get_state(TboBatch *b, uint32_t *words, uint8_t *done); /* float32 SoA words */
void tbo_get_state_f64(TboBatch *b, double *vals, uint8_t *done); /* same rows, full precision */
void tbo_set_state(TboBatch *b, const uint32_t *words, const uint8_t *done);

/* unit-level hooks for known-answer tests */
void tbo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int tbo_query_racket(const TbParams *p, const float rp[3], const float rq[4], const float c[3], double out[8]);
int tbo_query_racket_ground(const TbParams *p, const float rp[3], const float rq[4], double out[32]);
int tbo_get_manifold(TboBatch *b, int env, int32_t ids[4], double impulses[12]); /* racket<->court contact cache of one env */
int tbo_query_box(const TbParams *p, const float half[3], const float c[3], double out[4]);
int tbo_query_goal(const TbParams *p, float gx, float gy, const float c[3], double out[4]);

#ifdef __cplusplus
}
#endif
#endif
