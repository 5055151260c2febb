// Launchers of the backward-pass kernels (train_kernels.hip), shared with train.hip.
#pragma once
#include "kernels.h"

int omk_transpose(int dtype, const void* in, int64_t ldi, int64_t R, int C, void* out, int64_t ldo,
                  int64_t Rp, int op, hipStream_t s);
// n dense transposes out[i][c][r] = in[i][r][c] (R[i] x C[i]) in one launch per 40 matrices
int omk_transpose_batch(int dtype, const void* const* in, void* const* out, const int* R, const int* C, int n, hipStream_t s);
int omk_colsum(int dtype, const void* x, int64_t ld, int64_t M, int N, float* out, hipStream_t s);
int omk_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint64_t seed, hipStream_t s,
                const int* rows = nullptr /* packed rows: the token (b * L + position) of every row of width H -- the mask's key */, int H = 0);
int omk_ln_bwd(int dtype, const void* dy, const void* x, const float* g, void* dx, float* dg,
               float* db, int64_t M, int H, float eps, hipStream_t s);
// LayerNorm backward that also writes dx_drop = dropout(dx) (mask of the forward: seed, element index) when drop_p > 0
int omk_ln_bwd_drop(int dtype, const void* dy, const void* x, const float* g, void* dx, void* dx_drop, float drop_p,
                    uint64_t drop_seed, float* dg, float* db, int64_t M, int H, float eps, hipStream_t s,
                    const float* dy32 = nullptr /* the incoming gradient as f32 [M,H] instead of dy */,
                    const float* x32 = nullptr /* the normalisation's input as f32 [M,H] instead of x */,
                    float* partial = nullptr /* [OM_LNB_MAX_BLOCKS][2][H] f32: the blocks' column sums of d_gamma / d_beta go here (plain
                                                stores) instead of into dg / db (atomics); omk_ln_param_reduce adds them up */,
                    int* partial_blocks = nullptr /* out: how many blocks wrote */,
                    const int* drop_rows = nullptr /* packed rows: the token of every row (the dropout mask's key) */);
// the second half of that: dg[c] += sum_b partial[b][0][c], db likewise, for n sites in one launch, in a fixed order
#define OM_LNB_MAX_BLOCKS 512
#define OM_LN_SITES_MAX 32
struct OmLnSite { const float* partial; float* dg; float* db; int blocks; };
int omk_ln_param_reduce(const OmLnSite* sites, int n, int H, hipStream_t s);
// LayerNorm (rms = 0) or T5 RMSNorm (rms = 1) backward; `add` (optional, same shape) is added to dx
int omk_norm_bwd(int dtype, const void* dy, const void* x, const float* g, void* dx, float* dg,
                 float* db, int64_t M, int H, float eps, int rms, const void* add, hipStream_t s);
int omk_embed_bwd(int dtype, const void* dy, const int64_t* ids, const int64_t* type_ids,
                  const float* word, const float* pos, const float* type, const float* g,
                  float* dword, float* dpos, float* dtype_, float* dg, float* db, int64_t M, int L,
                  int H, int vocab, int type_vocab, float eps, hipStream_t s,
                  const int* cu = nullptr /* packed rows: dy holds sequence b at rows cu[b] .. cu[b + 1] - 1 (M stays B * L) */);
int omk_pool_bwd(int dtype, const float* dp, const int64_t* mask, void* dh, int64_t B, int L, int H,
                 int mode, hipStream_t s, const int* cu = nullptr /* packed rows: sequence b is rows cu[b] .. of dh */);
// rows [first[0], M) of a row-major tensor <- 0; `first` is on the device (packed rows: the pad rows behind the last sequence)
int omk_zero_rows_from(void* p, int64_t row_bytes, const int* first, int64_t M, hipStream_t s);
int omk_l2norm_bwd(const float* x, const float* dy, float* dx, int64_t M, int D, hipStream_t s);
int omk_small_nn(const float* A, const float* Bm, float* C, int I, int J, int Cc, hipStream_t s);
int omk_small_tn(const float* A, const float* Bm, float* C, int I, int J, int Cc, hipStream_t s);
int omk_attention_bwd(int dtype, const void* qkv, const void* dctx, void* dqkv, const int64_t* mask,
                      int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                      hipStream_t s, const int* cu = nullptr /* packed rows (16-bit, L <= 256): sequence b is rows cu[b] .. cu[b + 1] - 1 */);
int omk_attention_bwd_bias(int dtype, const void* qkv, const void* dctx, void* dqkv, const int64_t* mask,
                           int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                           const float* pos_bias, float* drel, hipStream_t s, const int* cu = nullptr /* packed rows, as above (the bias table keeps the pitch L) */);
// 256 < L <= 512, 16-bit formats (round 6): two kernels with one key / query tile in registers at a time; ctx = the forward's output (delta = dO . O)
int omk_attention_bwd_long(int dtype, const void* qkv, const void* ctx, const void* dctx, void* dqkv, const int64_t* mask,
                           int64_t B, int L, int H, int heads, float scale, float drop_p, uint64_t seed,
                           const float* pos_bias, float* drel, float* stats /* omk_attention_bwd_long_stats_bytes(B, heads) of scratch */, hipStream_t s);
size_t omk_attention_bwd_long_stats_bytes(int64_t B, int heads);
// bf16, L <= 128, no position bias: the transposing-read kernel of attention_bwd16.hip
bool omk_attention_bwd16_ok(int dtype, int L, int H, int heads);
int omk_attention_bwd16(int dtype, const void* qkv, const void* dctx, void* dqkv, const int64_t* mask, int64_t B, int L, int H,
                        int heads, float scale, float drop_p, uint64_t seed, hipStream_t s, const int* cu = nullptr,
                        const float* pos_bias = nullptr, float* drel = nullptr /* T5 (round 6): the bias table [heads][L][L] and its gradient per relative position */);
// T5 feed-forward activation (kind 0 relu, 1 gated gelu_new) forward / backward, embedding and bias backward
int omk_t5_act_fwd(int dtype, const void* f, const void* f2, void* g, int64_t n, int kind, hipStream_t s);
int omk_t5_act_bwd(int dtype, const void* dg, const void* f, const void* f2, void* df, void* df2, int64_t n, int kind, hipStream_t s);
int omk_t5_embed_bwd(int dtype, const void* dy, const int64_t* ids, float* dword, int64_t M, int H, int vocab, hipStream_t s,
                     const int* row_map = nullptr /* packed rows: row r carries token row_map[r] of ids (-1: a row no sequence owns) */);
int omk_t5_bias_bwd(const float* drel, const int* lut, float* dtable, int L, int heads, hipStream_t s);
