// Argument blocks of the rollout kernels (rollout.hip), shared with the C-ABI layer (capi.hip).
#pragma once
#include "common.hpp"

struct LstmStepArgs {
  const float* xproj; long ldx;      // row b at xproj + b * ldx (already offset to position t), 4*hs wide
  const float* h_prev;               // [B, hs]
  float* h_out;                      // [B, hs]  (must not alias h_prev: other workgroups still read it)
  float* c;                          // [B, hs]  updated in place
  const bf16_t* w_hh;                // [4*hs, hs]
  const int* lengths;                // [B] or null (all rows active)
  float* seq_out; long ld_seq;       // optional: row b at seq_out + b * ld_seq (already offset to position t), hs wide
  int B, hs, t;
  const int* xrow_start;             // optional (compacted xproj rows): row of (b, t) = xrow_start[b] + t, xproj then
  long ldx_row;                      // points at row 0 and ldx_row is the row stride (ldx unused)
};

struct SoftDotArgs {
  const float* target;          // [B, D]
  const float* context;         // [B, L, D], row stride ld_row, batch stride ld_batch (elements)
  long ld_batch, ld_row;
  const unsigned char* mask;    // [B, L] (nonzero = masked) or null
  float* weighted;              // [B, D] or null
  float* attn;                  // [B, L] or null: probabilities (output_prob) or the masked logits
  int B, L, D, output_prob;
};

struct SkinnyArgs {
  const float* x0; long ld0; int K0;   // [M, K0] fp32
  const float* x1; long ld1; int K1;   // [M, K1] fp32 or null: K-concatenated behind x0 (K0 % 4 == 0)
  const bf16_t* w; long ldw;           // [N, Kpad] bf16, zero past K0 + K1; Kpad % 32 == 0
  const float* bias;                   // [N] or null
  float* out; long ldo;                // [M, N] fp32
  int M, N, Kpad, act;                 // act: 0 none, 2 tanh
};
