// Internal launchers shared between translation units (not part of the C ABI).
#pragma once
#include "common.h"

int omk_layernorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* g,
                  const float* b, int64_t M, int H, float eps, int rms, hipStream_t s);
int omk_embed(int dtype, const int64_t* ids, const int64_t* type_ids, const float* word,
              const float* pos, const float* type, const float* g, const float* b, void* out,
              int64_t M, int L, int H, int vocab, int type_vocab, float eps, int bert,
              hipStream_t s);
int omk_pool(int dtype, const void* x, const int64_t* mask, float* out, int64_t B, int L, int H,
             int mode, hipStream_t s);
int omk_l2norm(const float* x, float* y, int64_t M, int D, hipStream_t s);
int omk_t5_bias(const float* table, const int* lut, float* out, int L, int heads, hipStream_t s);

// softmax(scale * Q K^T + mask [+ pos_bias]) V for every (batch, head); qkv is the fused
// projection output [B*L, 3H] (q | k | v), ctx is [B*L, H].  L <= 256, head_dim == 64.
int omk_attention(int dtype, const void* qkv, void* ctx, const int64_t* mask,
                  const float* pos_bias, int64_t B, int L, int H, int heads, float scale,
                  hipStream_t s);
