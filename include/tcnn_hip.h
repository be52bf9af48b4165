/*
 * tcnn_hip.h -- C ABI of the MI355X-native HashGrid + FullyFusedMLP hot path (libtcnn_hip.so).
 *
 * Plain C: opaque handles, raw device pointers, sizes, UTF-8 JSON text.  No C++/torch types cross
 * this boundary.  Every entry point names the reference interface it replaces (file:line under the
 * tiny-cuda-nn tree); INTEGRATION.md shows the reference-side bindings (pybind / C++ facade) that a
 * maintainer would point at these symbols.
 *
 * Conventions (identical to the reference's cpp_api, src/cpp_api.cu:72-153):
 *   - all matrices are "features x batch, column-major" == [batch][features] contiguous;
 *   - inputs are fp32, params / outputs / gradients are fp16 (TCNN_PRECISION_FP16);
 *   - n_elements must be a multiple of tcnn_batch_size_granularity() (256); the caller pads;
 *   - outputs have the PADDED width tcnn_module_n_output_dims() (multiple of 16); the caller slices;
 *   - dL_doutput arrives pre-multiplied by the loss scale (bindings/torch/tinycudann/modules.py:167);
 *   - the caller owns input/output/param/gradient buffers, the module owns only hyper-parameters;
 *     saved activations live in a tcnn_context_t the caller destroys.
 * Every function returns TCNN_OK (0) or an error code; tcnn_last_error() holds the message of the last
 * failure on the calling thread (the reference throws std::runtime_error, common_host.h:71-110).
 * Streams are hipStream_t passed as void*; NULL is the default stream.
 */
#ifndef TCNN_HIP_H
#define TCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCNN_OK 0
#define TCNN_ERROR 1             /* std::runtime_error in the reference */
#define TCNN_ERROR_UNSUPPORTED 2 /* part of the reference surface that is out of this build's scope */

typedef struct tcnn_module tcnn_module_t;                   /* tcnn::cpp::Module, cpp_api.h:91-119 */
typedef struct tcnn_context tcnn_context_t;                 /* tcnn::cpp::Context, cpp_api.h:87-89 */
typedef struct tcnn_trainable_model tcnn_trainable_model_t; /* tcnn::TrainableModel, config.h:46-51 */
typedef struct tcnn_train_context tcnn_train_context_t;     /* Trainer::ForwardContext, trainer.h:89-95 */
typedef void* tcnn_stream_t;                                /* hipStream_t (cudaStream_t in the reference) */

/* cpp_api.h:72-75.  BF16 is this build's extension: libtcnn_hip_bf16.so is the same library compiled with bfloat16 as
 * the parameter / activation / gradient type (the reference picks its type at compile time too, TCNN_HALF_PRECISION);
 * its tcnn_preferred_precision() / param_precision() / output_precision() report TCNN_PRECISION_BF16 and every `void*`
 * below that is documented as fp16 carries bfloat16 instead.  Snapshots name the type ("__half" / "__nv_bfloat16"). */
enum { TCNN_PRECISION_FP32 = 0, TCNN_PRECISION_FP16 = 1, TCNN_PRECISION_BF16 = 2 };
enum { TCNN_LOG_INFO = 0, TCNN_LOG_DEBUG, TCNN_LOG_WARNING, TCNN_LOG_ERROR, TCNN_LOG_SUCCESS }; /* cpp_api.h:52-58 */
enum { TCNN_GRADIENT_IGNORE = 0, TCNN_GRADIENT_OVERWRITE = 1, TCNN_GRADIENT_ACCUMULATE = 2 };  /* common.h:152-156 */

const char* tcnn_last_error(void);

/* ---- free functions, cpp_api.h:62-85 ------------------------------------------------------------- */
uint32_t tcnn_batch_size_granularity(void);            /* cpp_api.h:62 */
int tcnn_hip_device(void);                             /* cuda_device(), cpp_api.h:64 */
int tcnn_set_hip_device(int device);                   /* set_cuda_device(), cpp_api.h:65 */
void tcnn_free_temporary_memory(void);                 /* cpp_api.h:67 */
int tcnn_has_networks(void);                           /* cpp_api.h:69 */
float tcnn_default_loss_scale(int precision);          /* cpp_api.h:77 */
int tcnn_preferred_precision(void);                    /* cpp_api.h:79 */
int tcnn_supports_jit_fusion(int device);              /* cpp_api.h:81 -- always 0: no RTC path */
void tcnn_set_log_callback(void (*callback)(int severity, const char* message)); /* cpp_api.h:85 */

/* Device memory as the library allocates it (GPUMemory<T>, gpu_memory.h:97-130): hipMalloc / hipFree, or -- with the
 * environment variable TCNN_DEBUG_ALLOC=canary|fence set before the library is loaded -- a checking allocator
 * (csrc/device_alloc.h): poisoned blocks with canaries before and after (canary), or blocks that END on the last mapped
 * byte of their own mapping with unmapped address space on either side (fence: an out-of-bounds access faults at once).
 * tcnn_debug_alloc_mode: 0 off, 1 canary, 2 fence.  tcnn_debug_check_allocations: synchronises the device, verifies every
 * canary and returns the number of blocks that were written outside their bounds (details through tcnn_last_error()).
 * tcnn_set_debug_launches(1) (or TCNN_DEBUG_SYNC=1): every kernel launch is followed by a stream synchronisation and an
 * error check, and its name goes to stderr first when TCNN_DEBUG_TRACE=1 -- the last line names a faulting kernel. */
int tcnn_device_malloc(size_t bytes, void** out);
void tcnn_device_free(void* ptr);
int tcnn_debug_alloc_mode(void);
int tcnn_debug_check_allocations(void);
int tcnn_set_debug_launches(int enable);

/* ---- module factories, cpp_api.h:121-123 ---------------------------------------------------------- */
int tcnn_create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const char* encoding_json,
                                            const char* network_json, tcnn_module_t** out);
int tcnn_create_network(uint32_t n_input_dims, uint32_t n_output_dims, const char* network_json, tcnn_module_t** out);
int tcnn_create_encoding(uint32_t n_input_dims, const char* encoding_json, int requested_precision, tcnn_module_t** out);
void tcnn_module_destroy(tcnn_module_t* m);

/* ---- tcnn::cpp::Module methods, cpp_api.h:96-114 -------------------------------------------------- */
int tcnn_module_inference(tcnn_module_t* m, tcnn_stream_t stream, uint32_t n_elements, const float* input, void* output,
                          void* params);
int tcnn_module_forward(tcnn_module_t* m, tcnn_stream_t stream, uint32_t n_elements, const float* input, void* output,
                        void* params, int prepare_input_gradients, tcnn_context_t** ctx);
int tcnn_module_backward(tcnn_module_t* m, tcnn_stream_t stream, const tcnn_context_t* ctx, uint32_t n_elements,
                         float* dL_dinput, const void* dL_doutput, void* dL_dparams, const float* input,
                         const void* output, const void* params);
/* cpp_api.h:106-108 -> DifferentiableObject::backward_backward_input: second-order pass through dL_dinput.  Like the
 * reference, implemented by the grid encoding only (grid.h:910-1042); TCNN_ERROR_UNSUPPORTED for every other module.
 * dL_ddLdinput: fp32 [n][n_input_dims] (required); the three outputs dL_dparams (fp16, overwritten), dL_ddLdoutput (fp16
 * [n][padded width]) and dL_dinput (fp32, overwritten) are each optional; the context must come from a forward pass with
 * prepare_input_gradients. */
int tcnn_module_backward_backward_input(tcnn_module_t* m, tcnn_stream_t stream, const tcnn_context_t* ctx, uint32_t n_elements,
                                        const float* dL_ddLdinput, const float* input, const void* dL_doutput,
                                        void* dL_dparams, void* dL_ddLdoutput, float* dL_dinput, const void* params);
void tcnn_context_destroy(tcnn_context_t* ctx);

uint32_t tcnn_module_n_input_dims(const tcnn_module_t* m);
uint32_t tcnn_module_n_output_dims(const tcnn_module_t* m); /* PADDED width, cpp_api.cu:137 */
size_t tcnn_module_n_params(const tcnn_module_t* m);
int tcnn_module_param_precision(const tcnn_module_t* m);
int tcnn_module_output_precision(const tcnn_module_t* m);
int tcnn_module_initialize_params(tcnn_module_t* m, size_t seed, float* params_full_precision, float scale);
const char* tcnn_module_hyperparams_json(const tcnn_module_t* m); /* valid until the module is destroyed */
const char* tcnn_module_name(const tcnn_module_t* m);
int tcnn_module_jit_fusion(const tcnn_module_t* m);               /* always 0 */
int tcnn_module_set_jit_fusion(tcnn_module_t* m, int val);        /* accepted, ignored (warns if val != 0) */

/* Parity / debug helper (no reference counterpart): hash-grid entry index of every
 * (sample, level, corner) -> indices[(i * n_levels + level) * 2^D + corner], device uint32. */
int tcnn_module_grid_indices(tcnn_module_t* m, tcnn_stream_t stream, uint32_t n_elements, const float* input, uint32_t* indices);
/* Grid layout accessors used by the reference's own known-answer test (tests/test_grid.cu:55-71). */
int tcnn_module_grid_level_n_params(const tcnn_module_t* m, uint32_t level, size_t* out);
int tcnn_module_grid_level_params_offset(const tcnn_module_t* m, uint32_t level, size_t* out);

/* ---- create_from_config / Trainer / Network::inference (C++ template API) ------------------------ */
/* tcnn::create_from_config, config.h:53-63.  config_json holds "loss", "optimizer", "encoding", "network". */
int tcnn_create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const char* config_json, uint32_t seed,
                            tcnn_trainable_model_t** out);
void tcnn_trainable_model_destroy(tcnn_trainable_model_t* tm);

/* Trainer::training_step, trainer.h:254-357.  input: fp32 [batch][n_input_dims]; target: fp32
 * [batch][n_output_dims]; data_pdf (nullable) like target; dL_dinput (nullable) like input;
 * external_dL_dy (nullable): fp16 [batch][padded_output_width].  *ctx (nullable out) must be destroyed
 * with tcnn_train_context_destroy. */
int tcnn_trainer_training_step(tcnn_trainable_model_t* tm, tcnn_stream_t stream, uint32_t batch_size, const float* input,
                               const float* target, const float* data_pdf, int run_optimizer, float* dL_dinput,
                               int use_inference_params, int gradient_mode, const void* external_dL_dy,
                               tcnn_train_context_t** ctx);
/* Trainer::forward / backward / optimizer_step, trainer.h:97-157 */
int tcnn_trainer_forward(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale, uint32_t batch_size,
                         const float* input, const float* target, const float* data_pdf, int use_inference_params,
                         int prepare_input_gradients, const void* external_dL_dy, tcnn_train_context_t** ctx);
int tcnn_trainer_backward(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_train_context_t* ctx,
                          uint32_t batch_size, const float* input, float* dL_dinput, int use_inference_params,
                          int gradient_mode);
int tcnn_trainer_optimizer_step(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale);
/* Trainer::loss, trainer.h:372-374 (synchronises the stream) */
int tcnn_trainer_loss(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_train_context_t* ctx, float* out_loss);
void tcnn_train_context_destroy(tcnn_train_context_t* ctx);
/* ForwardContext members, trainer.h:89-95: fp16 [batch][padded_output_width] device pointers */
const void* tcnn_train_context_output(const tcnn_train_context_t* ctx);
const void* tcnn_train_context_dL_doutput(const tcnn_train_context_t* ctx);

/* DifferentiableObject::inference, object.h:214-271: fp32 in, fp32 [batch][n_output_dims] out (trimmed) */
int tcnn_network_inference(tcnn_trainable_model_t* tm, tcnn_stream_t stream, uint32_t batch_size, const float* input,
                           float* output, int use_inference_params);

/* parameter access, trainer.h:389-440 */
size_t tcnn_trainer_n_params(const tcnn_trainable_model_t* tm);
/* Mutable pointers (trainer.h:423-440 hands out T*): the library must assume the caller writes through them, now or later.  From the
 * first call on, the optimizer reads the 16-bit weights back instead of re-deriving skipped ones from its master weights, and the
 * transposed copy of the network weights is rebuilt before every pass -- a few per cent of a step -- until tcnn_trainer_params_written()
 * (after writes through _params / _params_inference: the 16-bit buffer is authoritative, the master weights are re-derived from it where
 * the two disagree) or tcnn_trainer_set_params_full_precision() (after writes to the master weights).  A host that only READS the master
 * weights (logging, checkpoints) uses tcnn_trainer_params_full_precision_view(): same memory, no change of mode. */
float* tcnn_trainer_params_full_precision(tcnn_trainable_model_t* tm);
const float* tcnn_trainer_params_full_precision_view(const tcnn_trainable_model_t* tm);
const void* tcnn_trainer_params_view(const tcnn_trainable_model_t* tm);   /* the 16-bit parameters, to read: no change of mode either */
void* tcnn_trainer_params(tcnn_trainable_model_t* tm);
void* tcnn_trainer_params_inference(tcnn_trainable_model_t* tm);
void* tcnn_trainer_param_gradients(tcnn_trainable_model_t* tm);
int tcnn_trainer_set_params_full_precision(tcnn_trainable_model_t* tm, const float* params, size_t n_params, int device_ptr);
int tcnn_trainer_set_params(tcnn_trainable_model_t* tm, const void* params_fp16, size_t n_params, int device_ptr);
/* Trainer::serialize / deserialize, trainer.h:442-481 (+ Adam::serialize, adam.h:304-325).  The snapshot is the
 * reference's JSON document {n_params, params_type, params_binary[, optimizer{...}]} as MessagePack bytes, i.e. what
 * nlohmann::json::to_msgpack(trainer->serialize(...)) yields.  Call with buffer == NULL to query *n_bytes. */
int tcnn_trainer_serialize(tcnn_trainable_model_t* tm, int serialize_optimizer, void* buffer, size_t capacity, size_t* n_bytes);
int tcnn_trainer_deserialize(tcnn_trainable_model_t* tm, const void* data, size_t n_bytes);
int tcnn_trainer_update_hyperparams(tcnn_trainable_model_t* tm, const char* json);      /* trainer.h:380-383 */
const char* tcnn_trainer_hyperparams_json(tcnn_trainable_model_t* tm);                   /* trainer.h:385-391 */
uint32_t tcnn_trainer_optimizer_step_count(const tcnn_trainable_model_t* tm);            /* Optimizer::step() */
uint32_t tcnn_trainer_padded_output_width(const tcnn_trainable_model_t* tm);
uint32_t tcnn_trainer_n_mlp_params(const tcnn_trainable_model_t* tm); /* "matrix" params: leading part of the buffer */

/* GPUMatrixDynamic<T> as it crosses the boundary (gpu_matrix.h:106-250): m rows (features) x n columns (samples);
 * column-major (CM == AoS, one sample's values contiguous, common.h:166-176): element (r, c) at data[c * stride + r];
 * row-major (RM == SoA): data[r * stride + c].  stride >= the leading dimension, in elements. */
#define TCNN_LAYOUT_ROW_MAJOR 0
#define TCNN_LAYOUT_COLUMN_MAJOR 1
typedef struct tcnn_matrix {
	void* data;
	uint32_t m, n, stride;
	int layout;
} tcnn_matrix_t;
/* Trainer::training_step / Network::inference with the reference's matrix-typed signatures (trainer.h:254-264,
 * object.h:214): `input`, `dL_dinput` and the inference `output` are GPUMatrixDynamic (any layout / stride); target,
 * data_pdf and external_dL_dy are GPUMatrix (dense column-major), as in the reference.  Optional arguments may be NULL. */
int tcnn_trainer_training_step_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_matrix_t* input, const tcnn_matrix_t* target,
                                        const tcnn_matrix_t* data_pdf, int run_optimizer, const tcnn_matrix_t* dL_dinput, int use_inference_params,
                                        int gradient_mode, const tcnn_matrix_t* external_dL_dy, tcnn_train_context_t** ctx_out);
int tcnn_network_inference_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_matrix_t* input, const tcnn_matrix_t* output,
                                    int use_inference_params);
/* Trainer::forward / Trainer::backward (trainer.h:97-148) the same way: `input` and `dL_dinput` of either layout. */
int tcnn_trainer_forward_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale, const tcnn_matrix_t* input, const tcnn_matrix_t* target,
                                  const tcnn_matrix_t* data_pdf, int use_inference_params, int prepare_input_gradients, const tcnn_matrix_t* external_dL_dy,
                                  tcnn_train_context_t** ctx_out);
int tcnn_trainer_backward_matrices(tcnn_trainable_model_t* tm, tcnn_stream_t stream, const tcnn_train_context_t* ctx, const tcnn_matrix_t* input,
                                   const tcnn_matrix_t* dL_dinput, int use_inference_params, int gradient_mode);

/* generate_random_uniform<float>(stream, rng, n, out, lower, upper) with `default_rng_t rng{seed}` (random.h:39-75,
 * pcg32.h:40-170): out[0..n) = U[lower, upper) from the pcg32 stream of `seed`, starting `*position` draws into it;
 * `*position` is advanced by n (what the reference's `rng.advance(n)` does), so successive calls continue one stream.
 * The synthetic inputs of samples/ and bench.py come from here (pcg32 seed 1337). */
int tcnn_generate_random_uniform(tcnn_stream_t stream, uint64_t seed, uint64_t* position, size_t n, float* out, float lower, float upper);

/* The regression target of the synthetic workloads (no reference counterpart in the library; the reference's sample evaluates its target
 * on the device inside the training loop, samples/mlp_learning_an_image.cu:263-271 `eval_image`): targets[i][c] = 0.5 + 0.5 sin(2 pi f x0)
 * cos(2 pi f x1) sin(2 pi x2 + c), f = c % 4 + 1, for sample-major positions [n][n_input_dims] -> [n][n_output_dims].  bench.py draws
 * a batch with tcnn_generate_random_uniform and evaluates this inside every timed step. */
int tcnn_generate_sinusoid_targets(tcnn_stream_t stream, uint32_t n, uint32_t n_input_dims, uint32_t n_output_dims, const float* positions, float* targets);

/* Data parallelism (no reference counterpart; the reference is single-GPU).
 * Loss gradients are normalised by global_batch_size * n_output_dims instead of the local batch, so the
 * SUM over ranks of the local gradient buffers equals the single-GPU gradient of the global batch.  The
 * host all-reduces tcnn_trainer_param_gradients() (RCCL) between backward and optimizer_step. */
/* tcnn_trainer_params / tcnn_trainer_params_inference expose a mutable pointer (trainer.h:489-503: the reference hands out its buffers
 * the same way); the library then rebuilds its transposed copy of the network weights before every pass.  Call this when done
 * writing through such a pointer: the copy is rebuilt once and trusted again (until the next tcnn_trainer_params call). */
int tcnn_trainer_params_written(tcnn_trainable_model_t* tm);
int tcnn_trainer_set_global_batch_size(tcnn_trainable_model_t* tm, uint64_t global_batch_size);
/* Optimizer step over the parameter range [begin, end) only (begin a multiple of 8).  Lets a data-parallel host step
 * each gradient bucket as soon as its all-reduce has finished, overlapping the optimizer with the remaining
 * communication.  One optimizer step == ranges that tile [0, n_params) exactly once, the range with begin == 0 first
 * (it advances the step counter and the learning-rate schedule). */
int tcnn_trainer_optimizer_step_range(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale, size_t begin, size_t end);
/* ONE optimizer step over the union of n_ranges parameter ranges only (begins multiples of 8): a rank that owns a shard of
 * the parameters (reduce-scatter of the gradients -> this -> all-gather of tcnn_trainer_params) steps just its shard; the
 * optimizer state of the other parameters is not touched on this rank. */
int tcnn_trainer_optimizer_step_ranges(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale, size_t n_ranges, const size_t* begins,
                                       const size_t* ends);
/* Gradient exchange hook for a data-parallel C/C++ host: training_step(run_optimizer = 1) calls `exchange` between backward
 * and the optimizer with the fp16 gradient buffer [network | encoding] on the step's stream; the host issues its
 * collective there, e.g. ncclAllReduce(g, g, n, ncclHalf, ncclSum, comm, (hipStream_t)stream).  The
 * library links no collective library itself.  exchange == NULL removes the hook. */
int tcnn_trainer_set_gradient_exchange(tcnn_trainable_model_t* tm, void (*exchange)(void* user, void* gradients_fp16, size_t n_params, tcnn_stream_t stream),
                                       void* user);
/* The reference's stream-ordered arena as a host sees it (GPUMemoryArena / allocate_workspace(stream, bytes), gpu_memory.h:405-700):
 * a block out of the library's stream-keyed cache (where its own scratch memory comes from).  tcnn_stream_free returns it to the cache --
 * not to the driver -- for later requests on the SAME stream (ordered behind its previous user); pass *granted back.  No device
 * synchronisation on either call in the steady state; tcnn_free_temporary_memory() releases the cache. */
int tcnn_stream_malloc(tcnn_stream_t stream, size_t bytes, void** out, size_t* granted);
int tcnn_stream_free(tcnn_stream_t stream, void* ptr, size_t granted);
/* Optimizer<T> on its own (reference optimizer.h:40-99, optimizers/adam.h): Adam over weight buffers the host owns.
 * create (JSON as in the "optimizer" block; otype Adam) -> allocate(n_weights, n_matrix_weights: the leading parameters that are matrix
 * weights, adam.h:79-110) -> step(stream, loss_scale, fp32 weights, 16-bit weights, 16-bit gradients scaled by loss_scale), any number of
 * times.  Same kernel and arithmetic as the trainer's optimizer step. */
typedef struct tcnn_optimizer tcnn_optimizer_t;
int tcnn_create_optimizer(const char* optimizer_json, tcnn_optimizer_t** out);
int tcnn_optimizer_allocate(tcnn_optimizer_t* o, size_t n_weights, size_t n_matrix_weights);
int tcnn_optimizer_step(tcnn_optimizer_t* o, tcnn_stream_t stream, float loss_scale, float* weights_full_precision, void* weights, const void* gradients);
uint32_t tcnn_optimizer_step_count(const tcnn_optimizer_t* o);
int tcnn_optimizer_update_hyperparams(tcnn_optimizer_t* o, const char* optimizer_json);
void* tcnn_optimizer_state(tcnn_optimizer_t* o, int which); /* 0 first moments, 1 second moments (fp32), 2 step counters (u32) */
void tcnn_optimizer_destroy(tcnn_optimizer_t* o);
/* Loss<T>::evaluate on its own (reference loss.h:42-50 and the headers under losses/): `loss_otype` as in the JSON ("RelativeL2", "L2", "L1", ...).
 * prediction / gradients: column-major `stride` x n matrices of the library's 16-bit type (stride = padded output width, a multiple of
 * 8), target / data_pdf (may be NULL): `dims` x n fp32, values (may be NULL): `stride` x n fp32.  Rows >= dims carry no loss; the
 * normalisation is n * dims as in the reference's kernels. */
int tcnn_loss_evaluate(const char* loss_otype, tcnn_stream_t stream, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale, const void* prediction,
                       const float* target, const float* data_pdf, float* values, void* gradients);
/* Exchange overlapped with the backward pass (no reference counterpart).  `ready(user, begin, end, stream)` is called on
 * the host, inside training_step, as soon as the kernels that produce the gradients [begin, end) of the fp16 gradient buffer have been
 * enqueued on `stream`: first the network's weights [0, n_network_params), then the encoding's levels in
 * tcnn_trainer_set_backward_level_groups() groups of consecutive levels (about equal parameter counts; default 1).  The ranges of one
 * step tile [0, n_params) in ascending order, begins are multiples of 8: a host starts that range's collective right there (behind an
 * event on `stream`) while the later groups are still being computed, and finishes the step with
 * tcnn_trainer_optimizer_step_range(s).  ready == NULL removes the hook. */
int tcnn_trainer_set_gradient_ready_callback(tcnn_trainable_model_t* tm, void (*ready)(void* user, size_t begin, size_t end, tcnn_stream_t stream), void* user);
int tcnn_trainer_set_backward_level_groups(tcnn_trainable_model_t* tm, uint32_t n_groups);
/* Data parallelism inside the library: `nccl_comm` is this rank's ncclComm_t (RCCL; NULL switches it off).  training_step then
 * all-reduces (sum) every ready range on an internal communication stream -- librccl.so is dlopen'ed by this call, the library does
 * not link it -- and, with run_optimizer = 1, steps each range as soon as ITS collective has finished while the later ones are still
 * on the wire; with run_optimizer = 0 the next tcnn_trainer_optimizer_step* call waits for them.  Set the global batch size
 * (tcnn_trainer_set_global_batch_size) so that the sum of the ranks' gradients is the global gradient. */
int tcnn_trainer_enable_rccl(tcnn_trainable_model_t* tm, void* nccl_comm, int n_ranks);
/* The same with the SHARDED exchange (ZeRO-1 style, what tinycudann/parallel.py's default does from Python): every ready range is
 * reduce-scattered (ncclReduceScatter, in place), training_step(run_optimizer = 1) runs Adam on this rank's shard of every range only and
 * all-gathers the 16-bit parameters (ncclAllGather, in place; the EMA weights too).  `rank` = this process's rank in the communicator.
 * Both schemes poll ncclCommGetAsyncError before they put the compute stream behind a collective: an asynchronous RCCL error (a dead peer)
 * surfaces as TCNN_ERROR with the communicator's message instead of a hang. */
int tcnn_trainer_enable_rccl_sharded(tcnn_trainable_model_t* tm, void* nccl_comm, int n_ranks, int rank);

/* Gradient exchange over PEER-MAPPED memory instead of ring collectives (csrc/direct_exchange.h; no reference counterpart).  On an MI355X
 * node every GPU has its own xGMI link to each peer: a rank that reads the P - 1 remote shards of ITS 1/P of the gradient buffer uses all
 * of its links at once (28.5 MB / 8 = 3.6 MB per link and phase at P = 8) where a ring pushes 7/8 of the buffer through one.
 *   tcnn_trainer_direct_export   fills `out` (n_bytes bytes; query the size with out == NULL) with IPC handles of this rank's trainer buffer
 *                                and signal block; the host passes every rank's record to every rank (any channel)
 *   tcnn_trainer_direct_open     maps the peers (exports: n_ranks records, rank r's at r * bytes_each); the HOST must put a barrier between
 *                                every rank's open and the first step
 *   training_step(run_optimizer = 1), or tcnn_trainer_direct_exchange_and_step after training_step(run_optimizer = 0):
 *                                signal + wait -> fp32 sum of this rank's shard over all ranks in rank order, ONE rounding -> Adam on the
 *                                shard (+ the < 8 P replicated tail parameters) -> stepped parameters written into every peer's buffer ->
 *                                signal + wait.  Loss gradients must be normalised by the global batch (tcnn_trainer_set_global_batch_size).
 *   tcnn_trainer_direct_status   *status = 0, or the phase (1 gradients, 2 parameters) in which a wait for the peers timed out
 *                                (TCNN_DIRECT_TIMEOUT_MS, default 2000: a dead peer yields an error, not a hung queue); synchronises
 * Not with Ema-wrapped optimizers, not under TCNN_DEBUG_ALLOC.  At most 16 ranks. */
int tcnn_trainer_direct_export(tcnn_trainable_model_t* tm, void* out, size_t capacity, size_t* n_bytes);
int tcnn_trainer_direct_open(tcnn_trainable_model_t* tm, int rank, int n_ranks, const void* exports, size_t bytes_each);
int tcnn_trainer_direct_close(tcnn_trainable_model_t* tm);
int tcnn_trainer_direct_exchange_and_step(tcnn_trainable_model_t* tm, tcnn_stream_t stream, float loss_scale);
int tcnn_trainer_direct_status(tcnn_trainable_model_t* tm, tcnn_stream_t stream, int* status);
/* link check of an opened exchange, to run once before it is trusted (collective: every rank, same rounds and seed, between steps; it
 * overwrites the gradient buffer and nothing else): rounds x {every rank fills its gradient buffer with a pattern, the shards are reduced
 * and pushed exactly as a step does it, every rank compares its whole buffer with the sum it must hold}.  *mismatches = elements of this
 * rank's buffer that were wrong (0 on a working node); *status as above. */
int tcnn_trainer_direct_selftest(tcnn_trainable_model_t* tm, tcnn_stream_t stream, uint32_t rounds, uint32_t seed, uint64_t* mismatches, int* status);
/* Adam's state (device pointers, n_params elements each): which = 0 first moments (fp32), 1 second moments (fp32),
 * 2 per-parameter step counters (u32; *steps_are_deficits = 1: the array holds `optimizer steps done - counter`). */
void* tcnn_trainer_optimizer_state(tcnn_trainable_model_t* tm, int which, int* steps_are_deficits);

/* Measurement hooks (no reference counterpart): HIP events recorded around each stage of the training step
 * on the stream the kernels are launched on.  only_stage < 0 times every stage, otherwise just that one.
 * tcnn_trainer_get_stage_times synchronises on the recorded events and returns accumulated totals. */
int tcnn_trainer_set_profiling(tcnn_trainable_model_t* tm, int enable, int only_stage);
int tcnn_trainer_n_stages(void);
const char* tcnn_trainer_stage_name(int stage);
int tcnn_trainer_get_stage_times(tcnn_trainable_model_t* tm, double* total_ms, uint64_t* counts);
/* Process-wide (default 1): training_step runs the network's forward +
 * loss + backward as one kernel, and forward() / backward() -- of a Trainer and of a module -- keep only the encoded input in
 * the context: the backward pass recomputes the hidden activations inside the same kernel instead of reading saved ones.
 * 0: separate forward (saving activations), loss and backward kernels.  Same results; contexts made under one setting must be
 * consumed under the same setting. */
int tcnn_get_fused_network_passes(void);
int tcnn_set_fused_network_passes(int enable);
/* training_step with an Identity encoding that pads nothing (n_input_dims a multiple of 16; identity.h:46-66): the network kernel reads the
 * caller's fp32 matrix itself where an instance offers it (training_step: 64 inputs, 64 neurons, one or two hidden layers -- the second is the
 * benchmarks/mlp shape; inference: every register-resident instance with 64 inputs) instead of running the encoding as a kernel of its own; same bits.  Process-wide, default on; 0 restores the separate kernel (A/B runs). */
int tcnn_set_fused_identity_input(int enable);
/* training_step(run_optimizer = 1) that runs the whole step itself (one GPU: no gradient exchange, no ready callback, GradientMode::Overwrite):
 * the fp32 weight-gradient slabs of the network kernel are summed by the first workgroups of the optimizer's launch -- the same additions in
 * the same order as the stand-alone finalize kernel, the same bits -- instead of by a launch of their own (fully_fused_mlp.cu:776-835 runs
 * split-K GEMMs there).  Process-wide, default on; 0 restores the separate kernel (A/B runs, tests). */
int tcnn_set_finalize_in_optimizer(int enable);
/* training_step as one graph launch (Trainer::training_step runs its passes under CudaGraph::capture_guard, trainer.h:343-350,
 * cuda_graph.h:65-155): with capture on, every training_step on a non-null stream that is not already capturing re-records its launches
 * into a graph, patches the instantiated graph with it (hipGraphExecUpdate; a changed topology re-instantiates) and launches that.  The first
 * step of a shape (batch size, which optional arguments are present) runs plainly so that the stream's scratch cache holds every block the
 * capture will ask for; steps with a gradient exchange, a ready callback, a direct exchange or profiling run plainly too.  Default OFF: on
 * MI355X a replayed step measures the same as the plain one at every batch size, the switch exists for hosts that want
 * the reference's behaviour.  _stats: graph launches and instantiations so far. */
int tcnn_trainer_set_graph_capture(tcnn_trainable_model_t* tm, int enable);
int tcnn_trainer_graph_capture_stats(const tcnn_trainable_model_t* tm, uint64_t* launches, uint64_t* instantiations);
/* Tuning knob: bytes of LDS one grid-backward workgroup uses for the table slice it owns (default 64 KiB). */
int tcnn_trainer_set_lds_level_budget(tcnn_trainable_model_t* tm, uint32_t bytes);
/* Grid backward formulation, process-wide: 0 = owner-computes LDS slices with fp32 accumulation on hashed levels,
 * 1 = same with packed-fp16 accumulation (default; the reference's accumulation type), 2 = the reference's
 * per-corner global atomics (A/B measurements), 3 = bucket-once: large levels derive each corner once, bin the
 * records by owning slice in HBM queues and accumulate them exactly (64-bit fixed point) in the owner's LDS.
 * Also selectable with TCNN_GRID_BACKWARD=sliced_f32|sliced_f16|atomic|bucketed. */
int tcnn_set_grid_backward_mode(int mode);
int tcnn_get_grid_backward_mode(void);
/* Accumulator form of the bucket owners in mode 3, process-wide (same bits from all three): 0 = packed (default: both features of a
 * payload word in one 64-bit LDS word, half the LDS atomics and half the LDS; slices whose sums could leave int32 are redone wide),
 * 1 = 64-bit fixed point per value throughout, 2 = the packed kernel with every slice through its wide redo (tests). */
int tcnn_set_grid_owner_mode(int mode);
int tcnn_get_grid_owner_mode(void);
/* Table slices the packed owners had to redo with 64 bits per value since the process started (their gradients failed the int32 bound:
 * sum of |gradient| over the slice >= 120); each redo costs that slice twice the time.  Synchronises the device. */
int tcnn_grid_owner_wide_slices(uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* TCNN_HIP_H */
