/* orc_shade.h — internal declarations shared by orc_shade.c and orc_wavefront.c.
 * TEST INFRASTRUCTURE ONLY (see nexus_oracle.h). */
#ifndef ORC_SHADE_H
#define ORC_SHADE_H

#include "nexus_oracle.h"
#include "orc_math.h"

f3 orc_cosine_hemisphere(uint32_t *rng);
f2 orc_unit_disk(uint32_t *rng);
int orc_pdf_valid(float pdf);
float orc_power_heuristic(float a, float b);
uint32_t orc_uniform(uint32_t max, uint32_t *rng);
f2 orc_uniform_triangle(uint32_t *rng);
int orc_bsdf_sample_f3(const nx_material *m, f3 wi, uint32_t *rng, f3 *wo, f3 *thr, float *pdf);
int orc_bsdf_eval_f3(const nx_material *m, f3 wi, f3 wo, f3 *thr, float *pdf);
float orc_srgb_to_linear(uint8_t c);
f4 orc_tex2d_f4(const nx_texture_desc *t, float u, float v);

int orc_trace_one(const orc_scene *s, const float org[3], const float dir[3], int anyHit, float tmaxIn, nx_hit *hit,
                  orc_trace_stats *st);

#endif
