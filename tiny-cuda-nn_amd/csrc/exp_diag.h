// exp_diag.h -- the experiment switches of the kernels: timing ladders that skip ONE phase of a kernel so that its stage time says what
// the phase costs (profiles/r03_exp_notes.txt section 10, r04_exp_notes.txt).  The results of such a build are WRONG ON PURPOSE.
//
// In the product build this header is empty of effect: the switches are compile-time zeros, the code they guard is dead (the ISA of the
// shipped kernels is the same with and without the guarded lines).  An experiment build passes -DTCNN_EXPERIMENT together with a switch
// (scripts/build_variant_one.sh adds the flag for every -DTCNN_EXP_* it is given):
//   * a switch without the flag is a compile error;
//   * csrc/Makefile, which builds the shipped libraries, refuses the flag;
//   * an object built with it exports `tcnn_experiment_build_marker`: tests/test_library.py requires the shipped libraries to have no
//     such symbol, tinycudann._C.is_experiment_build() tells a process which kind it loaded and bench.py says so in its line.
#pragma once
#include <cstdint>

#if defined(TCNN_EXPERIMENT)
extern "C" __attribute__((weak, visibility("default"))) int tcnn_experiment_build_marker = 1;
#else
#if defined(TCNN_EXP_DIAG_SCATTER) || defined(TCNN_EXP_DIAG_OWNER) || defined(TCNN_EXP_FWD_REGION_LOG2)
#error "TCNN_EXP_* switches exist in experiment builds only: add -DTCNN_EXPERIMENT (scripts/build_variant_one.sh does)"
#endif
#endif

namespace tcnn_hip {
// k_grid_bucket_scatter: bit 0 no queue stores, bit 1 no reservation atomics, bit 2 no reordering stores, bit 3 no rank atomics
#if defined(TCNN_EXP_DIAG_SCATTER)
constexpr uint32_t EXP_DIAG_SCATTER = TCNN_EXP_DIAG_SCATTER;
#else
constexpr uint32_t EXP_DIAG_SCATTER = 0u;
#endif
// k_grid_bucket_owner: bit 0 no table clear, bit 1 no LDS atomics, bit 2 no conversion / store, bit 3 no last-owner protocol
#if defined(TCNN_EXP_DIAG_OWNER)
constexpr uint32_t EXP_DIAG_OWNER = TCNN_EXP_DIAG_OWNER;
#else
constexpr uint32_t EXP_DIAG_OWNER = 0u;
#endif
// k_grid_forward_tiles, region passes: a hashed level whose table has more than 2^TCNN_EXP_FWD_REGION_LOG2 entries is
// walked once per slice of that many entries; every pass fetches only the corners that fall inside its slice (the other lanes' loads are
// masked off), so that an XCD works on a slice its L2 can hold.  A TIMING build: every pass overwrites the level's features with its own
// partial sums (an exact version would have to stage the raw corners -- the fp16 fma chain of grid.h:144-163 is order-dependent).
// TCNN_EXP_FWD_REGION_COST: what the work plan charges a (slice, tile) item, in the units of make_forward_plan (a 2 MiB hashed level: 8).
#if defined(TCNN_EXP_FWD_REGION_LOG2)
constexpr uint32_t EXP_FWD_REGION_LOG2 = TCNN_EXP_FWD_REGION_LOG2;
#else
constexpr uint32_t EXP_FWD_REGION_LOG2 = 0u;
#endif
#if defined(TCNN_EXP_FWD_REGION_COST)
constexpr double EXP_FWD_REGION_COST = TCNN_EXP_FWD_REGION_COST;
#else
constexpr double EXP_FWD_REGION_COST = 4.0;
#endif
}  // namespace tcnn_hip
