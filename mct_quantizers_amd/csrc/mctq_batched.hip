// mctq_batched.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h): a LIST of affine fake-quantizations in one launch
// (mctq_fq_batched, mctq_fq_batch_pack / _run, and the one-tensor gather launch of mctq_fq_per_channel).  Body: mctq_batched.hpp.
#define MCTQ_BATCHED_PART 1
#include "mctq_batched.hpp"
