/* oracle.h -- TEST INFRASTRUCTURE, not product code.
 * C interface of liboracle.so, the CPU restatement of the reference's scalar_rgb path / volpath
 * (see oracle.cpp).  Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg only. */
#ifndef ORACLE_H
#define ORACLE_H
#include "../include/mtsamd.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct oracle_scene oracle_scene;
const char *oracle_last_error(void);
int oracle_scene_create(const mts_scene_desc *desc, oracle_scene **out);
int oracle_scene_destroy(oracle_scene *s);
int oracle_cancel(oracle_scene *s);
int oracle_render(oracle_scene *s, int n_threads, int shard_index, int shard_count, float *film, mts_stats *stats);
int oracle_sample(oracle_scene *s, int32_t n, uint64_t seed_offset, const float *ox, const float *oy, const float *oz,
                  const float *dx, const float *dy, const float *dz, float *out_rgb, uint8_t *out_valid);
int oracle_ray_intersect(oracle_scene *s, int32_t n, const float *o, const float *d, const float *mint, const float *maxt,
                         float *out_t, int32_t *out_shape, int32_t *out_prim, float *out_p, float *out_n);
#ifdef __cplusplus
}
#endif
#endif
