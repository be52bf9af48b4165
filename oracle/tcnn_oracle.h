/*
 * tcnn_oracle.h -- CPU restatement of the tiny-cuda-nn HashGrid + FullyFusedMLP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (tiny-cuda-nn_amd/, include/)
 * may include, link or call this.  Allowed users: tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg.
 *
 * Parity pinning status (see DESIGN.md "Oracle"; tests/test_oracle_ref.py):
 *   - PINNED, bit for bit, against the reference's own code compiled for the host (oracle/_ref/libtcnn_ref.so, built by
 *     oracle/build_ref.py from the sources where they lie under /root/reference; manifest.json lists file:lines):
 *       grid_scale / grid_resolution / grid_index / pos_fract        common_device.h:767-895, 1000-1043
 *       kernel_grid (encoding and dy_dx), kernel_grid_backward (records; sums to the error of the reference's own running
 *       fp16 sum), kernel_grid_backward_input                       encodings/grid.h:48-349
 *       adam_step                                                    optimizers/adam.h:47-127
 *       the element-wise losses                                      losses/{l2,relative_l2,l1,relative_l1,mape,smape,relative_l2_luminance,cross_entropy,variance_is}.h
 *       pcg32 + generate_random_kernel                               random.h:39-69, dependencies/pcg32/pcg32.h
 *       warp_activation / warp_activation_backward                   common_device.h:108-186, 363-440 (up to the sign of a zero)
 *       the identity encoding                                        encodings/identity.h:45-85
 *       GridEncodingTemplated's constructor: level sizes, offset table, n_params (host code, compiled inside a stand-in for its class)
 *                                                                    encodings/grid.h:673-737
 *       FullyFusedMLP::initialize_params + GPUMatrix::initialize_xavier_uniform: matrices, draw order, ranges, generator state left
 *       behind (likewise)                                            src/fully_fused_mlp.cu:868-893, gpu_matrix.h:292-307
 *       the second-order grid kernels (dL_ddLdy bit for bit; dL_dx and the grid gradient to the rounding of the reference's own
 *       running atomic sums)                                         encodings/grid.h:351-653
 *       frequency_encoding[_backward], kernel_one_blob_soa, kernel_one_blob_backward   encodings/frequency.h:45-105, oneblob.h:98-164
 *     and against the reference's known answers for the grid layout (/root/reference/tests/test_grid.cu:55-71) and its literal
 *     constants (common_device.h:787-791, 854-866).  The committed fixture tests/golden/reference_small.npz was produced by that
 *     library; the HIP path is held against it on the GPU without the oracle in the loop.
 *     What "the reference" means there: its source, every float operation rounded as written (-ffp-contract=off).  nvcc contracts
 *     a * b + c into one fma by default; where the oracle models that (the uniform transform of random.h:63) it says so.
 *       kernel_mlp_fused / kernel_mlp_fused_backward and the threadblock_* device functions they call
 *                                                                    src/fully_fused_mlp.cu:46-557
 *       -- the fully fused network kernels, compiled against oracle/ref_shim/mma.h (nvcuda::wmma for the host) with a thread block's
 *       threads run as fibers (oracle/ref_driver_mlp.cpp).  This file's fp16-accumulate mode reproduces them BIT FOR BIT: hidden
 *       activations, padded output, inference output, backward activations' end result dL/dinput, for widths 16-128, 1-5 hidden layers,
 *       all eight activations, both matrix layouts.  What is modelled there and not the reference's: the arithmetic INSIDE one 16x16x16
 *       tensor-core operation (exact products, binary32 sum in ascending k, one rounding to the binary16 accumulator) -- the PTX ISA
 *       leaves the hardware's internal order open, so no host build could do better.
 *   - PARITY UNPINNED, and only this: the CUTLASS GEMMs (weight gradients fully_fused_mlp.cu:776, 819, 829; output layers wider than
 *     16; dL/dinput of inputs narrower than the network; src/cutlass_mlp.cu) -- the CUTLASS submodule is not in /root/reference.  They
 *     are plain matrix products of buffers that ARE pinned (the fused kernels' forward and backward activations): this file restates
 *     them with fp16 storage and either accumulator type (fp32: what MFMA does, the default; fp16: cutlass_matmul.h:67), tests hold the
 *     sums against float64 products of the reference kernel's own activations, and the GPU result must land between the two modes
 *     (tests/test_oracle.py, tests/test_oracle_ref.py, tests/test_gpu_parity_full.py).
 *
 * Conventions: "half" values travel as uint16_t bit patterns (IEEE binary16, RNE conversions).
 * Matrices named AoS are [N][width] (one sample's features contiguous == the reference's
 * column-major features x batch GPUMatrix, common.h:166-176).
 */
#ifndef TCNN_ORACLE_H
#define TCNN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 128 /* multi_level_interface.h:84-88 MAX_N_LEVELS */
#define ORC_MAX_DIMS 4

enum { ORC_GRID_HASH = 0, ORC_GRID_DENSE = 1, ORC_GRID_TILED = 2 };
enum { ORC_INTERP_NEAREST = 0, ORC_INTERP_LINEAR = 1, ORC_INTERP_SMOOTHSTEP = 2 };
enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_LEAKY_RELU = 2, ORC_ACT_EXPONENTIAL = 3, ORC_ACT_SIGMOID = 4, ORC_ACT_SQUAREPLUS = 5,
       ORC_ACT_SOFTPLUS = 6, ORC_ACT_TANH = 7 }; /* common_device.h:108-186, 363-418 */
enum { ORC_LOSS_L2 = 0, ORC_LOSS_RELATIVE_L2 = 1, ORC_LOSS_L1 = 2, ORC_LOSS_RELATIVE_L1 = 3, ORC_LOSS_MAPE = 4, ORC_LOSS_SMAPE = 5,
       ORC_LOSS_CROSS_ENTROPY = 6, ORC_LOSS_VARIANCE = 7, ORC_LOSS_RELATIVE_L2_LUMINANCE = 8 }; /* names: src/loss.cu:57-65 */

/* ---- the 16-bit type: IEEE fp16 (default) or bfloat16 (the product's -DTCNN_BF16 build); process-wide switch ---- */
void orc_set_half_format(int bf16);
int orc_get_half_format(void);
uint16_t orc_f2h(float f);
float orc_h2f(uint16_t h);
void orc_f2h_array(const float* in, uint16_t* out, size_t n);
void orc_h2f_array(const uint16_t* in, float* out, size_t n);

/* ---- pcg32 (dependencies/pcg32/pcg32.h:40-170) ---- */
typedef struct {
	uint64_t state;
	uint64_t inc;
} orc_pcg32;
void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq);
uint32_t orc_pcg32_next_uint(orc_pcg32* r);
float orc_pcg32_next_float(orc_pcg32* r);
void orc_pcg32_advance(orc_pcg32* r, int64_t delta);
/* std::seed_seq{seed}.generate(2 words), first word (trainer.h:53-56) */
uint32_t orc_seed_seq_first(uint32_t seed);
/* random.h:39-75: generate_random_uniform with the kernel's idx <-> stream-position mapping */
void orc_generate_random_uniform(orc_pcg32* rng, size_t n, float* out, float lower, float upper);

/* ---- grid encoding ---- */
typedef struct {
	uint32_t n_dims;
	uint32_t n_levels;
	uint32_t n_features_per_level;
	uint32_t log2_hashmap_size;
	uint32_t base_resolution;
	float per_level_scale;
	int grid_type;
	int interpolation;
	uint32_t offsets[ORC_MAX_LEVELS + 1]; /* in grid entries (not features) */
	float scale[ORC_MAX_LEVELS];
	uint32_t resolution[ORC_MAX_LEVELS];
	uint32_t n_params; /* offsets[n_levels] * F */
} orc_grid;

/* grid.h:673-737.  returns 0 on success */
int orc_grid_init(orc_grid* g, uint32_t n_dims, uint32_t n_levels, uint32_t n_features_per_level,
                  uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale,
                  int grid_type, int interpolation);

/* common_device.h:847-884 grid_index for one corner */
uint32_t orc_grid_index(const orc_grid* g, uint32_t level, const uint32_t* pos_grid);

/* Per (sample, level, corner) entry index [N][L][2^D] and fp32 weight; for bit-exact tests */
void orc_grid_indices(const orc_grid* g, const float* positions /*[N][D]*/, uint32_t n,
                      uint32_t* indices /*[N][L][2^D]*/, float* weights /*[N][L][2^D] or NULL*/);

/* grid.h:49-212 forward.  params: half bits [n_params].  out: half bits AoS [N][out_stride],
 * columns >= L*F are zero (grid.h:757-766).  fp16 fma chain exactly as grid.h:144-163.
 * dy_dx (optional): fp32 [N][L*F][D] (grid.h:172-211). */
void orc_grid_forward(const orc_grid* g, const uint16_t* params, const float* positions, uint32_t n,
                      uint16_t* out, uint32_t out_stride, float* dy_dx);

/* grid.h:215-320 backward.  dL_dy: half bits AoS [N][dy_stride].  grad: double [n_params],
 * accumulated (+=) with each contribution rounded to half as the reference does
 * ((GRAD_T)weight * grad, grid.h:254) but summed exactly -- the "ideal" atomics result. */
void orc_grid_backward_stochastic(const orc_grid* g, const float* positions, uint32_t n, const uint16_t* dL_dy,
                                  uint32_t dy_stride, double* grad);
void orc_grid_backward(const orc_grid* g, const float* positions, uint32_t n, const uint16_t* dL_dy,
                       uint32_t dy_stride, double* grad);

/* grid.h:323-349 input gradient: dL_dx [N][D] fp32 */
/* second order (grid.h:352-655): see the definition; every output may be NULL */
void orc_grid_backward_backward_input(const orc_grid* g, const uint16_t* params, const float* positions, const float* ddx, uint32_t n,
                                      const uint16_t* dL_dy, uint32_t dy_stride, const float* dy_dx, double* grad_params,
                                      uint16_t* dL_ddLdy, float* dL_dx);
void orc_grid_backward_input(const orc_grid* g, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride,
                             const float* dy_dx, float* dL_dx);
/* GridEncodingTemplated<float> (Encoding<float>, cpp_api.cu:165-168): fp32 parameters / features / gradients, fp32 fma interpolation;
 * layouts as above; orc_grid_backward_f32 ADDS the exact (double) sums of the fp32 products into `grad`. */
void orc_grid_forward_f32(const orc_grid* g, const float* params, const float* positions, uint32_t n, float* out, uint32_t out_stride, float* dy_dx);
void orc_grid_backward_f32(const orc_grid* g, const float* positions, uint32_t n, const float* dL_dy, uint32_t dy_stride, double* grad);
void orc_grid_backward_input_f32(const orc_grid* g, uint32_t n, const float* dL_dy, uint32_t dy_stride, const float* dy_dx, float* dL_dx);

/* ---- MLP (cutlass_mlp.cu:162-316, fully_fused_mlp.cu:635-678 param layout) ---- */
typedef struct {
	uint32_t in_width;      /* padded input width (multiple of 16) */
	uint32_t width;         /* hidden width */
	uint32_t out_width;     /* real output width */
	uint32_t padded_out;    /* next multiple of 16 */
	uint32_t n_hidden;      /* n_hidden_layers >= 1 */
	int activation;
	int output_activation;
	uint32_t n_params;
} orc_mlp;

int orc_mlp_init(orc_mlp* m, uint32_t in_width, uint32_t width, uint32_t out_width, uint32_t n_hidden,
                 int activation, int output_activation);
/* fully_fused_mlp.cu:868-893 + gpu_matrix.h:292-307 (Xavier uniform, row-major draw order) */
void orc_mlp_init_params(const orc_mlp* m, orc_pcg32* rng, float* params_fp32, float scale);

/* forward: input half AoS [N][in_width]; hidden (optional, may be NULL) half [n_hidden][N][width]
 * post-activation; output half AoS [N][padded_out].  accum_fp16 != 0 emulates half accumulators
 * rounded every 16 k-steps (cutlass_matmul.h:67-68) -- used only to bracket the reference's error. */
void orc_activation_forward(int act, uint32_t n, const uint16_t* x, uint16_t* y);
void orc_activation_backward(int act, uint32_t n, const uint16_t* v, const uint16_t* y, uint16_t* out);
void orc_mlp_forward(const orc_mlp* m, const uint16_t* params, const uint16_t* input, uint32_t n,
                     uint16_t* hidden, uint16_t* output, int accum_fp16);

/* backward: dL_doutput half [N][padded_out]; needs input + hidden from forward.
 * grad_params: double [n_params] (+=, caller zeroes for Overwrite) -- may be NULL;
 * dL_dinput half [N][in_width] -- may be NULL. */
void orc_mlp_backward(const orc_mlp* m, const uint16_t* params, const uint16_t* input, const uint16_t* hidden,
                      const uint16_t* output, const uint16_t* dL_doutput, uint32_t n, double* grad_params,
                      uint16_t* dL_dinput);

/* the same with the reference's half accumulators emulated (see the definition) */
void orc_mlp_backward_ex(const orc_mlp* m, const uint16_t* params, const uint16_t* input, const uint16_t* hidden,
                         const uint16_t* output, const uint16_t* dL_doutput, uint32_t n, double* grad_params,
                         uint16_t* dL_dinput, int accum_fp16);

/* ---- losses (losses/relative_l2.h:40-76, losses/l2.h:40-76) ----
 * prediction half [N][stride]; target fp32 [N][dims]; values fp32 [N][stride]; gradients half [N][stride].
 * n_total_override: 0 -> N*dims (reference); else used as n_total (data-parallel global batch). */
void orc_loss(int loss_type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale,
              const uint16_t* prediction, const float* target, const float* data_pdf, float* values,
              uint16_t* gradients, uint64_t n_total_override);

/* ---- Adam (optimizers/adam.h:48-127) ---- */
typedef struct {
	float learning_rate, beta1, beta2, epsilon, l2_reg, non_matrix_l2_reg;
	float relative_weight_decay, absolute_weight_decay;
	float weight_clipping_magnitude, gradient_clipping_magnitude;
	float non_matrix_learning_rate_factor;
	int adabound;
	int optimize_matrix_params, optimize_non_matrix_params, skip_zero_grad_non_matrix_params;
} orc_adam_hparams;
void orc_adam_defaults(orc_adam_hparams* h);
/* gradients: half bits; current_step is the optimizer-wide step AFTER increment (adam.h:159) */
void orc_adam_step(const orc_adam_hparams* h, uint32_t n, uint32_t n_matrix_weights, float loss_scale,
                   uint32_t current_step, float* weights_fp32, uint16_t* weights_half,
                   const uint16_t* gradients, float* m1, float* m2, uint32_t* param_steps);

/* ---- frequency encoding (encodings/frequency.h:46-104): in [N][n_dims] fp32, out / dL_dy [N][padded] half, pads with 1.0 ---- */
void orc_frequency_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint16_t* out);
void orc_frequency_backward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, const uint16_t* dL_dy, float* dL_dx);

/* ---- one-blob encoding (encodings/oneblob.h:84-164): in [N][n_dims] fp32, out / dL_dy [N][padded] half, pads with 1.0 ---- */
void orc_oneblob_forward(uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint16_t* out);
void orc_oneblob_backward(uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, const uint16_t* dL_dy, float* dL_dx);

/* ---- identity encoding (encodings/identity.h:46-66): pads with 1.0 ---- */
void orc_identity_forward(uint32_t n, uint32_t n_dims, uint32_t padded, const float* in /*[N][n_dims]*/,
                          uint16_t* out /*[N][padded]*/);

/* ---- whole training step of HashGrid+MLP (trainer.h:254-357), used as cpu_baseline ("port") ----
 * All buffers owned by caller.  grads_half receives the half-rounded gradient buffer
 * [mlp | grid] (trainer.h:489-495 layout).  Returns summed loss. */
typedef struct {
	orc_grid grid;
	orc_mlp mlp;
	int loss_type;
	orc_adam_hparams adam;
	uint32_t n_params;    /* mlp.n_params + grid.n_params */
	uint32_t n_out;       /* real output dims */
} orc_model;
int orc_model_init(orc_model* md, uint32_t n_in, uint32_t n_out, const orc_grid* g, uint32_t width,
                   uint32_t n_hidden, int loss_type, const orc_adam_hparams* adam);
double orc_training_step(const orc_model* md, uint32_t n, const float* positions, const float* targets,
                         float* params_fp32, uint16_t* params_half, uint16_t* grads_half, float* m1,
                         float* m2, uint32_t* steps, uint32_t current_step, float loss_scale,
                         int run_optimizer, uint16_t* out_prediction /*[N][padded_out] or NULL*/);
/* optional arguments of Trainer::training_step (trainer.h:254-264); any pointer may be NULL */
typedef struct {
	const float* data_pdf;          /* [N][n_out] */
	const uint16_t* external_dL_dy; /* half [N][padded_out], replaces the loss gradient (trainer.h:124-128) */
	float* dL_dinput;               /* out: [N][n_dims] */
	int accum_fp16;                 /* network GEMMs with the reference's half accumulators */
	uint64_t n_total_override;      /* loss normalisation count of the GLOBAL batch under data parallelism, 0 = N*dims */
} orc_step_options;
double orc_training_step_ex(const orc_model* md, uint32_t n, const float* positions, const float* targets,
                            float* params_fp32, uint16_t* params_half, uint16_t* grads_half, float* m1,
                            float* m2, uint32_t* steps, uint32_t current_step, float loss_scale,
                            int run_optimizer, uint16_t* out_prediction, const orc_step_options* opt);
void orc_inference(const orc_model* md, uint32_t n, const float* positions, const uint16_t* params_half,
                   float* out /*[N][n_out]*/);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
