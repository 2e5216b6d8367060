// Caller-owned library state (ABI 10; SURVEY 8(b): no process-global state behind the C ABI).
// Everything the entry points used to keep in file-scope globals lives in a D3dContext: the d3d_set_tuning knobs (test /
// sweep hooks) and the ticket words of the deterministic cross-block folds (d3d_fold.h), one ring per device.  A context
// is created by the caller (d3d_ctx_create), bound to a host thread (d3d_ctx_bind) and used by every entry point that
// thread calls afterwards; threads that bind nothing share the process default context, whose knobs start from the
// D3D_* environment variables -- which is exactly the behaviour of ABI <= 9.  Two host threads (two streams) with a
// context each never share a knob or a ticket word.
#pragma once
#include <atomic>

struct D3dKnobs {
  int fwd_variant;       // 0: tiling variant of the sparse forward kernel (-1 = D3D_FWD_VARIANT)
  int wgrad_blocks;      // 1
  int xcd_remap;         // 2
  int bn_variant;        // 3
  int bn_skip_global;    // 4 (ignored)
  int bn_dx_rows;        // 5
  int bn_stream_blocks;  // 6
  int c2_variant;        // 7
  int c2_wgrad_waves;    // 8
  int c2_tail;           // 9
  int sp_exact_f32;      // 10 (test knob)
  int fwd_resident;      // 11
};
#define D3D_FOLD_RING_WORDS (1 << 16)
#define D3D_MAX_DEVICES 16
struct D3dContext {
  D3dKnobs k;
  unsigned *ring[D3D_MAX_DEVICES];
  std::atomic<unsigned> next[D3D_MAX_DEVICES];
  std::atomic_flag lock;
};
// the calling thread's bound context, else the process default
D3dContext *d3d_ctx_cur();

// the knob names the kernels' launchers were written with
#define g_fwd_variant (d3d_ctx_cur()->k.fwd_variant)
#define g_d3d_wgrad_blocks (d3d_ctx_cur()->k.wgrad_blocks)
#define g_xcd_remap (d3d_ctx_cur()->k.xcd_remap)
#define g_bn_variant (d3d_ctx_cur()->k.bn_variant)
#define g_bn_skip_global (d3d_ctx_cur()->k.bn_skip_global)
#define g_bn_dx_rows (d3d_ctx_cur()->k.bn_dx_rows)
#define g_bn_stream_blocks (d3d_ctx_cur()->k.bn_stream_blocks)
#define g_c2_variant (d3d_ctx_cur()->k.c2_variant)
#define g_c2_wgrad_waves (d3d_ctx_cur()->k.c2_wgrad_waves)
#define g_c2_tail (d3d_ctx_cur()->k.c2_tail)
#define g_sp_exact_f32 (d3d_ctx_cur()->k.sp_exact_f32)
#define g_fwd_resident (d3d_ctx_cur()->k.fwd_resident)
