// Persistent decode chain (decode_chain.hip): host-side entry points used by cover_decoder_forward.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cover_hip.h"

bool decode_chain_supported(const cover_dec_desc* d, int rows);
size_t decode_chain_ws_bytes();
int decode_chain_status();
// stage 0: [sums of squares(x) -> norm -> qkv(L)];  stage 1: [o_proj(L) -> gate_up(L) -> down(L) (-> qkv(next))]
hipError_t launch_decode_chain(const cover_dec_desc* d, int stage, const cover_dec_layer* L, const cover_dec_layer* next, void* x, void* qkv,
                               void* attn, void* mlp, float* ssq, int rows, bool split, hipStream_t st);
