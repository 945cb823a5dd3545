// ems_kernels.h -- launchers of the gfx950 kernels (ems_kernels.hip), called by ems_core.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ems_plan.h"

int ems_lpe_for(int dc);                                     // lanes per edge row; 0 = unsupported degree
int ems_max_waves_for(int dc, int q);                      // waves per workgroup k_decode is built for
size_t ems_wave_lds_bytes(int q, int dc, int nm, int n_lists);
size_t ems_wg_lds_bytes(int q, int M, size_t wave_bytes, int waves);
bool ems_has_q16t_kernel(const EmsPlan &P);                 // GF(16) dc 8 nm 8 in lane-serial form (k_decode_q16t)
size_t ems_q16t_wg_lds_bytes(int n_groups, int waves);
int ems_q16t_max_waves();
hipError_t ems_launch_q16t_export(const EmsPlan &P, int B, hipStream_t st);   // a decode on k_decode_q16t's travelling layout back to app / ctov
int ems_wide_max_waves(int q);                                 // waves per workgroup k_decode_wide is built for
int ems_wide_lpe(int q);                                     // lanes per edge row of k_decode_wide (check degree above 16)
size_t ems_wide_wave_lds_bytes(int q, int dc, int nm, int nbuf);   // nbuf: list buffers per wave (EmsPlan::wide_nbuf)
bool ems_has_occ4_kernel(const EmsPlan &P);
bool ems_has_dual_kernel(const EmsPlan &P);                  // two check-node visits per wave (GF(256) dc 4 nm 20; GF(16) dc 8 nm 8)
size_t ems_v2wide_wave_lds_bytes(int q, int dc, int nm, int nbuf);    // one wave of k_decode_v2_wide (the v2 flavour on any check degree; P.t_dc set)
bool ems_has_v2_dual_kernel(const EmsPlan &P);               // the v2 flavour's two-visit kernel (GF(256) dc 4 nm 20)
bool ems_has_pair16_kernel(const EmsPlan &P);                // two visits per wave at sixteen waves per CU (GF(256) dc 4 nm 20; k_decode_pair16)
size_t ems_pair16_wave_lds_bytes(int q, int nm);
size_t ems_pair16_wg_lds_bytes(int q, int nm, int n_pairs, int waves);
bool ems_has_twin_kernel(const EmsPlan &P);                  // two FRAMES per wave, check nodes in reference order (tiny GF(64) dc 4 codes, nm 16)
int ems_dual_max_waves(int dc);                              // waves per workgroup k_decode_dual is built for
size_t ems_dual_wave_lds_bytes(int q, int dc, int nm);
size_t ems_dual_wg_lds_bytes(int q, int dc, int nm, int n_pairs, int waves);
hipError_t ems_launch_v2_init(const float *like, const int *like_idx, int B, int N, int q, int nm, int E, const int *vn_edge,
                              float *likeS, uint8_t *msg, int rec, int *bad, hipStream_t st);
hipError_t ems_launch_decode_v2(const EmsPlan &P, int B, hipStream_t st, const char **kname = nullptr);
hipError_t ems_launch_v2_init_rows(const float *likeS, int B, int N, int q, int nm, int E, const int *vn_edge, uint8_t *msg, int rec,
                                   int *bad, hipStream_t st);
hipError_t ems_launch_sort_like_desc(const float *likeS, int N, int q, float *s_like, int *s_idx, hipStream_t st);

// kname (optional) receives the launched kernel's name, template arguments included: a static string
hipError_t ems_launch_decode(const EmsPlan &P, int B, hipStream_t st, const char **kname = nullptr);
hipError_t ems_launch_llr_init(const float *noisy_dev, float sigma, int logq, int q, size_t total,
                               const uint8_t *idx2bin_dev, const uint8_t *bin2idx_dev, float *app_dev, hipStream_t st);
hipError_t ems_launch_llr_const(const float *obs_dev, const float *att_dev, const float *table_dev, int q, int D, int mode, float sigma,
                                size_t nvn, const uint8_t *bin2idx_dev, float *app_dev, hipStream_t st);
hipError_t ems_launch_count_errors(const uint8_t *decide_dev, const int *cw_dev, int B, int N, int K,
                                   const uint8_t *idx2bin_dev, int *nberr_dev, hipStream_t st);
hipError_t ems_launch_expand_ctov(const uint8_t *rec_dev, int rec_bytes, int nm, int q, int E, float *out_dev,
                                  hipStream_t st);
hipError_t ems_launch_es_unit(const float *in1, const float *in2, const int *idx1, const int *idx2, float *out,
                              int *idxout, int nm, int nbOper, int q, const uint8_t *idx2bin, const uint8_t *bin2idx,
                              hipStream_t st);
hipError_t ems_launch_cn_unit(const EmsPlan &P, const int *coef, const float *vtoc_llr, const int *vtoc_gf, float *ctov,
                              hipStream_t st);
hipError_t ems_launch_decide_rows(const float *app_dev, const int *rows_dev, int nrows, int B, int N, int q, uint8_t *decide_dev, hipStream_t st);
hipError_t ems_launch_decision_unit(const float *app, int N, int q, int *decide, hipStream_t st);
hipError_t ems_launch_syndrome_unit(const int *row_ptr, const int *col, const int *coef, int M, int q, const int *decide,
                                    const uint8_t *idx2bin, const uint8_t *bin2idx, unsigned *first_bad, hipStream_t st);
hipError_t ems_launch_sort_llr(const float *app_dev, int N, int q, float *s_llr_dev, int *s_gf_dev, hipStream_t st);
hipError_t ems_launch_source(const EmsSourcePlan &S, uint64_t seed, uint64_t first_frame, int B, float sigma, int *cw_dev,
                             float *noisy_dev, hipStream_t st);
// the reference's drand48 stream on the device (ems_source.hip): base_dev[f] = LCG state at frame f's first draw, pow_a / pow_c = the LCG's
// 2^i-step coefficients; samples whose float rounding the device cannot decide are listed in flags (count in flag_count, may exceed cap)
hipError_t ems_launch_source_reference(const EmsSourcePlan &S, const uint64_t *base_dev, const uint64_t pow_a[24], const uint64_t pow_c[24], int B, float sigma,
                                       int *cw_dev, float *noisy_dev, unsigned *flag_count_dev, void *flags_dev, unsigned flag_cap, hipStream_t st);
hipError_t ems_launch_source_patch(const void *flags_dev, const float *vals_dev, unsigned n, int N, int logq, float *noisy_dev, hipStream_t st);
hipError_t ems_debug_counters(unsigned long long *out8, int reset); // diagnostic builds (-DEMS_STAMPS) only
