// Launch batcher (host side, C++): between dav_batch_begin() and dav_batch_end() the library's entry points RECORD
// their kernel launches instead of issuing them.  Recorded launches are organised in LANES — sequences of launches
// the caller declares mutually independent (the image tower block and the audio tower block of one layer,
// models/deepavfusion.py:104-105; the two MAE decoders, models/avmae.py:147-180; the two aggregation cross-attentions
// of the fusion block, models/fusion_blocks.py:240-243).  dav_batch_end() walks the lanes in lockstep: the k-th
// launches of all lanes are independent of each other, so those of one kernel family and tile configuration are
// issued as ONE grouped grid (problem table in the kernel arguments, each workgroup looks its problem up), the rest
// one by one.  Order inside a lane is preserved, so the result is the one sequential execution gives.
//
// Why: on MI355X the step is a chain of ~1900 dependent launches whose individual grids (150-900 tiles of 128x128)
// fill 512 workgroup slots badly (1.4 rounds = 28 % quantisation) and hipGraph branches of equal weight do not
// overlap (rocprofv3 r01: image and audio qkv GEMMs overlap 0.0 us of 900).  Grouping makes each grid 2-3x larger
// and halves the length of the dependent chain.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>

namespace davb {

// launches n >= 1 recorded parameter blocks of one family + configuration (n == 1: the plain kernel)
typedef void (*GroupFn)(const void* const* params, int n, hipStream_t stream);

bool recording();
void push_opaque(std::function<void()> fn);                                    // any other launch: replayed as is
void push_typed(GroupFn fn, const void* params, size_t bytes, hipStream_t stream);

}  // namespace davb
