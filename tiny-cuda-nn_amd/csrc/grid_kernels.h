// grid_kernels.h -- multiresolution hash/dense/tiled grid encoding on gfx950.
//
// Restates the BEHAVIOUR of reference include/tiny-cuda-nn/encodings/grid.h:49-349 (kernel_grid,
// kernel_grid_backward, kernel_grid_backward_input) and common_device.h:767-895,1000-1043 with an
// MI355X-first mapping:
//   * one (level, sample-tile) work item per workgroup, and workgroup -> level chosen so that the
//     workgroups an XCD receives (block b runs on XCD b%8) all walk the same one or two levels: a
//     level's table (<= 2^log2_hashmap_size * F * 2 B, 2 MiB at the headline config) then stays in
//     that XCD's private 4 MiB L2 while it is being gathered / scattered.  Placement only changes
//     speed, never results.
//   * the per-level scale/resolution come from a host-computed table (bit-exact indices on any
//     device), the interpolation is the reference's fp16 fma chain (v_pk_fma_f16) -> bit-exact
//     encodings versus the CPU oracle.
//   * backward: packed-half atomics (global_atomic_pk_add_f16) for large levels; levels whose whole
//     table fits in LDS are accumulated per workgroup in fp32 LDS (ds_add_f32) and flushed once.
#pragma once
#include "tcnn_device.h"

namespace tcnn_hip {

struct GridMeta {
	uint32_t n_dims;      // D
	uint32_t n_levels;    // L
	uint32_t n_feat;      // F  (features per level)
	uint32_t grid_type;   // GridType
	uint32_t interp;      // InterpolationType
	float max_level;      // reference MultiLevelEncoding::m_max_level (1.0 = all levels)
	uint32_t stochastic;  // grid.h:284-299: backward sends (sample, level)'s gradient to ONE randomly chosen corner (forward is unaffected)
	uint32_t offset[MAX_N_LEVELS + 1];
	float scale[MAX_N_LEVELS];
	uint32_t resolution[MAX_N_LEVELS];
};

struct GridIO {
	// positions: element (dim d, sample i) at positions[i * pos_stride_i + d * pos_stride_d]
	const float* positions;
	uint32_t pos_stride_i, pos_stride_d;
	uint32_t n;  // samples
	// encoded output / dL_dy: element (feature k, sample i) at ptr[k * stride_k + i * stride_i]
	//   SoA (feature-major, the reference's preferred layout grid.h:1070-1072): stride_k = n, stride_i = 1
	//   AoS (cpp_api.cu:94-95 forces it):                                       stride_k = 1, stride_i = padded width
	uint32_t stride_k, stride_i;
	// second-order pass only: dL/d(dL_dx), element (dim d, sample i) at ddx[i * ddx_stride_i + d * ddx_stride_d]
	const float* ddx = nullptr;
	uint32_t ddx_stride_i = 0, ddx_stride_d = 0;
};

// Forward: writes n_levels*F features (padding columns are the caller's job). dy_dx may be null;
// layout [(k * n + i) * D + d] fp32 (grid.h:784, 209).
void grid_forward(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* params, half_t* out,
                  float* dy_dx);


// Backward into grid_gradient (half).  accumulate == false overwrites (GradientMode::Overwrite: any zeroing
// the chosen mode needs is done here, the caller does not memset), true adds to what is there.
//   SlicedF32 / SlicedF16: owner-computes LDS accumulation (fp32, or packed fp16 like the reference's own
//                          half2 atomics) -- no global atomics on large levels; the default.
//   Atomic:                the reference's formulation, global_atomic_pk_add_f16 per corner (F >= 2 only);
//                          kept for A/B measurements.
//   Bucketed:              large levels derive every corner ONCE, bin the records by owning slice in HBM queues
//                          and let the owner accumulate them in 64-bit fixed point in LDS; small levels as in
//                          the sliced modes.  Needs a GridBackwardWorkspace (sizes from grid_backward_workspace_size):
//                          `scratch` is device memory the call may scribble on (nothing is kept in it between
//                          calls); `counters` must be ZERO on entry and is left zeroed by the call, so a caller
//                          keeps one such buffer per stream and never clears it again.
// lds_slice_bytes: LDS bytes one workgroup devotes to its table slice (0 = default 128 KiB).
enum class GridBackwardMode : int { SlicedF32 = 0, SlicedF16 = 1, Atomic = 2, Bucketed = 3 };
struct GridBackwardWorkspace {
	void* scratch = nullptr;
	size_t scratch_bytes = 0;
	uint32_t* counters = nullptr;
	size_t n_counters = 0;
	// optional: called right before (begin = 1) and after (begin = 0) each kernel sequence of the backward --
	// phase 0: record scatter (bucketed mode only), 1: accumulation + stores (every mode)
	void (*phase_hook)(void* user, int phase, int begin) = nullptr;
	void* hook_user = nullptr;
};
GridBackwardWorkspace grid_backward_workspace_size(const GridMeta& meta, uint32_t n, GridBackwardMode mode, uint32_t lds_slice_bytes);
void grid_backward(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, half_t* grid_gradient, bool accumulate,
                   GridBackwardMode mode, uint32_t lds_slice_bytes, const GridBackwardWorkspace& workspace = GridBackwardWorkspace());

// Accumulator form of the bucket owners (pass B of the Bucketed mode), process-wide (tcnn_set_grid_owner_mode):
//   0 packed  -- the two features of a payload word share one 64-bit LDS word (one ds_add_u64 per table entry at F = 2, 8 bytes of
//                LDS per entry, two workgroups per CU); a slice whose gradients could leave int32 is redone with 64 bits per value
//   1 fixed64 -- 64 bits per value throughout (k_grid_backward_sliced; also what odd F runs)
//   2 wide    -- the packed kernel, every slice through its 64-bit redo (tests of that path)
// All three produce the same bits.
int& grid_owner_mode();
// Slices the packed owners redid with 64 bits per value since the process started (synchronises the device): each costs that slice
// twice the time -- a workload whose gradients keep failing the int32 bound (sum of |gradient| over a slice >= 120) shows up here.
unsigned long long grid_owner_wide_slices();

// dL_dx[i][d] = sum_k dL_dy[k][i] * dy_dx[k][i][d]   (grid.h:323-349)
void grid_backward_input(hipStream_t stream, uint32_t n_dims, uint32_t n_features, const GridIO& io,
                         const half_t* dL_dy, const float* dy_dx, float* dL_dx, uint32_t dx_stride_i,
                         uint32_t dx_stride_d);

// ---- fp32 encodings: GridEncodingTemplated<float> (cpp_api.cu:165-168).  Parameters / encoded features / gradients fp32, the reference's
// formulation (fp32 fma interpolation, one fp32 global atomic per corner and feature); same GridIO addressing as above.
void grid_forward_f32(hipStream_t stream, const GridMeta& meta, const GridIO& io, const float* params, float* out, float* dy_dx = nullptr);
void grid_backward_f32(hipStream_t stream, const GridMeta& meta, const GridIO& io, const float* dL_dy, float* grid_gradient, bool accumulate);
void grid_backward_input_f32(hipStream_t stream, uint32_t n_dims, uint32_t n_features, const GridIO& io, const float* dL_dy, const float* dy_dx, float* dL_dx,
                             uint32_t dx_stride_i, uint32_t dx_stride_d);

// ---- second order: gradients of dL_dx = sum_k dL_dy[k] * dy_dx[k] (the first backward's input gradient) --------------
// (grid.h:352-655, 910-1042: what SDF / eikonal losses differentiate through)
//  * w.r.t. the grid parameters: grid_backward() with io.ddx set -- the same scatter with the corner weight
//    scale * sum_d ddx[d] * pos'(d) * (+-1 along d) * prod_{e != d} w_e  (kernel_grid_backward_input_backward_grid);
//  * w.r.t. dL_dy:  dL_ddLdy[k][i] = sum_d dy_dx[k][i][d] * ddx[i][d]   (kernel_grid_backward_input_backward_dLdoutput);
//    padding features k in [n_features, n_features + n_to_pad) are written as zero;
//  * w.r.t. the positions (kernel_grid_backward_input_backward_input): mixed second derivatives of the interpolation
//    (and the diagonal ones for Smoothstep); dL_dx[i][d] is OVERWRITTEN.  io.ddx must be set for all three.
void grid_backward_backward_dLdoutput(hipStream_t stream, uint32_t n_dims, uint32_t n_features, uint32_t n_to_pad, const GridIO& io,
                                      const float* dy_dx, half_t* dL_ddLdy);
void grid_backward_backward_input(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, const half_t* params,
                                  float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_d);

// Debug/parity helper: entry index per (sample, level, corner) -> indices[(i*L + l)*2^D + c]
void grid_indices(hipStream_t stream, const GridMeta& meta, const GridIO& io, uint32_t* indices);

}  // namespace tcnn_hip
