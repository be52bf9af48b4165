// direct_exchange.h -- gradient exchange of a data-parallel step over PEER-MAPPED memory instead of ring collectives.
//
// Why: on an MI355X node every GPU has a direct xGMI link to each of its 7 peers (~153 GB/s per link and direction).
// A ring reduce-scatter + all-gather pushes (P - 1) / P of the 28.5 MB gradient buffer twice through ONE outgoing link per rank
// (>= 0.33 ms per step at P = 8 against 0.24 ms of forward + backward); reading the P - 1 remote shards of one's OWN 1/P of the buffer
// directly moves 1/P of the buffer over EACH of the 7 links at the same time: 28.5 MB / 8 = 3.6 MB per link and phase, ~23 us at link
// rate, twice per step.
//
// Scheme (one process per GPU; every rank maps its peers' trainer buffers with hipIpcOpenMemHandle):
//   1. after its backward pass a rank tells every peer "my gradients of step s are final" (a 4-byte system-scope store into the peer's
//      signal block) and waits until it has heard the same from all of them;
//   2. k_direct_reduce: the rank reads ITS shard of every rank's gradient buffer (own memory + P - 1 peers), adds them in fp32 IN RANK
//      ORDER and rounds once to the 16-bit type -- deterministic, the same sum on whichever rank computes it, one rounding instead of the
//      P - 1 a ring's partial sums go through -- and writes the result into its own gradient buffer; the < 8 P parameters that do not
//      divide belong to the LAST rank's shard (shards need not be equal): every parameter is reduced, stepped and pushed by ONE rank;
//   3. Adam on the rank's shard (tcnn_trainer_optimizer_step_ranges: the optimizer shrinks by P, as in the sharded collective scheme);
//   4. k_direct_push: the rank writes its stepped 16-bit parameters into every peer's parameter buffer, signals "pushed s", and waits for
//      everybody's push before the next forward pass reads the parameters (which also tells it that nobody still reads its gradients).
// Signals are step counters (monotonic, never reset); a wait gives up after a timeout (TCNN_DIRECT_TIMEOUT_MS, default 2000) and records
// an error instead of hanging the queue (direct_exchange_status()); the error word is copied to pinned host memory behind every step
// and the next step's exchange throws when it is set, so that a starved wait cannot desynchronise the replicas silently.
// No collective library is involved.
//
// Status: exercised with two and four ranks sharing ONE GPU (tests/test_gpu_distributed.py): handles, signals, the reduction and the
// push are real; the links are not.  Unmeasured on a multi-GPU node.
#pragma once
#include "tcnn_device.h"

#include <algorithm>
#include <vector>

namespace tcnn_hip {

constexpr int DIRECT_MAX_RANKS = 16;
constexpr size_t DIRECT_HANDLE_BYTES = 64;  // sizeof(hipIpcMemHandle_t)

// what a rank publishes to its peers (plain bytes: travels through any host-side channel)
struct DirectExport {
	unsigned char buffer_handle[DIRECT_HANDLE_BYTES];  // the trainer's [master | params | grads] allocation
	unsigned char signal_handle[DIRECT_HANDLE_BYTES];  // its signal block
	uint64_t params_offset, grads_offset;              // bytes from the start of the buffer allocation
	uint64_t n_params;
};

struct DirectExchange {
	int rank = -1, n_ranks = 0;
	uint64_t n_params = 0;
	size_t shard = 0;                      // parameters per rank (a multiple of 8); the last rank's shard also holds the remainder
	size_t own_begin = 0, own_end = 0;     // this rank's shard [own_begin, own_end): every parameter has exactly ONE owner
	size_t own_count() const { return own_end - own_begin; }
	uint32_t blocks() const { return (uint32_t)std::min<size_t>((own_count() / 8 + 255) / 256 + 1, 2048); }
	half_t* grads[DIRECT_MAX_RANKS] = {};   // every rank's gradient buffer as mapped HERE (own entry: the trainer's own pointer)
	half_t* params[DIRECT_MAX_RANKS] = {};
	uint32_t* signals[DIRECT_MAX_RANKS] = {};  // [2][DIRECT_MAX_RANKS] words per rank: row 0 "gradients of step s final", row 1 "parameters of step s pushed"
	uint32_t* own_signals = nullptr;           // this rank's block (allocated here, exported)
	uint32_t* error_flag = nullptr;            // device word: non-zero once a wait timed out
	volatile uint32_t* host_error = nullptr;   // pinned host copy of it, refreshed behind every step's push: the NEXT step fails instead of training on
	void* step_done = nullptr;                 // hipEvent_t recorded behind that refresh: begin_step waits for the step before, so the copy it reads is that step's
	bool step_recorded = false;
	void* mapped_buffers[DIRECT_MAX_RANKS] = {};  // bases returned by hipIpcOpenMemHandle (to close)
	void* mapped_signals[DIRECT_MAX_RANKS] = {};
	uint32_t step = 0;
	uint32_t timeout_ms = 2000;
	bool active() const { return n_ranks > 0; }
};

// this rank's signal block + error word (idempotent); fills `out` for the trainer buffer at `buffer` (base of its hipMalloc allocation)
void direct_exchange_export(DirectExchange& dx, void* buffer, const half_t* params, const half_t* grads, uint64_t n_params, DirectExport& out);
// maps the peers; exports[r] is what rank r published (exports[rank] must be this rank's own)
void direct_exchange_open(DirectExchange& dx, int rank, int n_ranks, const DirectExport* exports, half_t* own_params, half_t* own_grads);
void direct_exchange_close(DirectExchange& dx);
// the phases of a step one by one (what direct_exchange_reduce / _push are made of): begin_step checks the error word of the steps before and
// advances the step counter; signal_wait(row 0) = "my gradients are final" + wait for everybody's; reduce_own; [the caller's Adam on
// [own_begin, own_end)]; push_own; signal_wait(row 1) = "pushed" + wait; finish_step refreshes the host copy of the error word
void direct_exchange_begin_step(DirectExchange& dx);
void direct_exchange_signal_wait(hipStream_t stream, DirectExchange& dx, int row);
void direct_exchange_reduce_own(hipStream_t stream, DirectExchange& dx);
void direct_exchange_push_own(hipStream_t stream, DirectExchange& dx);
void direct_exchange_finish_step(hipStream_t stream, DirectExchange& dx);
// phase 1 + 2 of a step: signal, wait, reduce this rank's shard into the own gradient buffer
void direct_exchange_reduce(hipStream_t stream, DirectExchange& dx);
// phase 4: push the own parameter shard to every peer, signal, wait
void direct_exchange_push(hipStream_t stream, DirectExchange& dx);
// link check (collective; clobbers the gradient buffers): rounds x {pattern -> reduce -> push the reduced shards -> compare the whole buffer}
void direct_exchange_selftest(hipStream_t stream, DirectExchange& dx, uint32_t rounds, uint32_t seed, uint64_t* mismatches, int* status);
// 0 while every wait found its signals in time (synchronises the stream)
int direct_exchange_status(hipStream_t stream, DirectExchange& dx);

}  // namespace tcnn_hip
