// primitives.h -- the three data-parallel primitives of the set-up passes (device_analysis.hip, device_ordering.hip), hand-written for
// gfx950 in primitives.hip: exclusive scan, stable LSD radix sort, runs of a sorted key array.  Rounds 3 - 5 took them from rocPRIM: 898
// kernel instantiations and 8.8 MB of code object in a 9.2 MB library for a dozen calls per uploaded graph, loaded by the first upload of
// every process.  All launches are stream-ordered on `s`; `scratch` is caller-provided device memory of at least the stated size.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace gt {
namespace prim {

// out[i] = in[0] + ... + in[i - 1], i = 0 .. n - 1 (sums in int64)
size_t scan_scratch_bytes(size_t n);
void exclusive_scan(const int64_t* in, int64_t* out, size_t n, void* scratch, hipStream_t s);
void exclusive_scan(const int32_t* in, int64_t* out, size_t n, void* scratch, hipStream_t s);

// Stable sort by the low `bits` bits of the keys (least-significant-digit radix sort, 8-bit digits: equal keys keep their order -- the
// order of the terms inside a Schur block, and of the factors inside an incidence list, is the input order).  The sorted arrays arrive
// in key_out / val_out; key_in / val_in are used as the second buffer of the passes and hold garbage afterwards.
size_t sort_scratch_bytes(size_t n);
void sort_pairs(uint64_t* key_in, uint64_t* key_out, uint32_t* val_in, uint32_t* val_out, size_t n, int bits, void* scratch, hipStream_t s);
void sort_pairs(uint32_t* key_in, uint32_t* key_out, uint32_t* val_in, uint32_t* val_out, size_t n, int bits, void* scratch, hipStream_t s);
void sort_keys(uint64_t* key_in, uint64_t* key_out, size_t n, int bits, void* scratch, hipStream_t s);

// The runs of equal keys of a SORTED array: uniq[r] = the key of run r, start[r] = its first position, start[n_runs] = n (start may be
// null: the distinct keys only), *n_runs (device) = their number.  uniq needs room for n keys, start for n + 1 offsets.
size_t runs_scratch_bytes(size_t n);
void runs(const uint64_t* sorted, size_t n, uint64_t* uniq, int64_t* start, int32_t* n_runs, void* scratch, hipStream_t s);

}  // namespace prim
}  // namespace gt
