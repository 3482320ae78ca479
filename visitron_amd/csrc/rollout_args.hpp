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
  // training (all optional, rows of an inactive position are left as the caller initialised them):
  float* sv_gates; long ld_svg;      // activated gates i, f, g, o of this step: row b at sv_gates + b * ld_svg, 4*hs wide
  float* sv_c; long ld_svc;          // the cell state BEFORE this step, hs wide
  bf16_t* sv_h; long ld_svh;         // the hidden state BEFORE this step as bf16 (the weight gradient's operand), hs wide
};

// One step of back-propagation through time for the same recurrence (lstm_step_bwd_kernel).
struct LstmBwdArgs {
  const bf16_t* dg_next; long ld_dgn;   // gate gradients of the step that consumed this step's h (row b at + b * ld), or null
  const bf16_t* w_hh_t;                 // W_hh transposed: [hs, 4*hs] bf16
  const float* dh_final;                // [B, hs] or null (zeros): gradient of the final hidden state
  const float* d_out; long ld_dout;     // gradient of the padded output at this position (row b at + b * ld) or null
  float* dc;                            // [B, hs] running cell-state gradient (initialised to the final state's), in place
  const float* sv_gates; long ld_svg;   // what the forward step saved
  const float* sv_c; long ld_svc;
  bf16_t* dg_out; long ld_dg;           // this step's gate gradients (pre-activation), 4*hs wide, zeros for inactive rows
  float* dg_out_f32; long ld_dgf;       // optional fp32 copy (the cell's dense backward reads it)
  const int* lengths;                   // [B] or null
  int B, hs, t, t_next;                 // t_next: position of dg_next (a row inactive there takes dh_final instead)
};

struct SoftDotBwdArgs {
  const float* target;          // [B, D]
  const float* context;         // [B, L, D] (strides as the forward)
  long ld_batch, ld_row;
  const unsigned char* mask;    // [B, L] or null
  const float* d_weighted;      // [B, D] or null
  const float* d_attn;          // [B, L] or null: gradient of the returned probabilities (output_prob) / masked logits
  float* d_target;              // [B, D]
  float* d_context;             // [B, L, D] contiguous, or null
  int B, L, D, output_prob;
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
