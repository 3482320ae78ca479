// Argument block shared by the weight-gradient kernels (gemm_wgrad.hip, gemm_wgrad_v8.hip) and capi.hip.
#pragma once
#include "common.hpp"

#define WG_MAX_PROBLEMS 8

struct WgradProblem {
  const bf16_t* dY; long ldy;   // [M, N]
  const bf16_t* X; long ldx;    // [M, K]
  float* dW; long ldw;          // [N, K] fp32
  float* db;                    // [N] fp32 or null
  int N, K;
  int tiles_k;                  // ceil(K / 128)
  int tile_begin;               // first linear tile id of this problem
  int accumulate;               // 0: overwrite dW/db, 1: add to them
};

struct WgradArgs {
  WgradProblem p[WG_MAX_PROBLEMS];
  int nprob;
  int M;
};

