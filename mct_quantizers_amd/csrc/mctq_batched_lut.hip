// mctq_batched_lut.hip -- part of libmctq_hip.so (C ABI: include/mctq_hip.h): a LIST of decision-table LUT quantizations in
// one launch (mctq_lutt_batch_pack / _run).  Body: mctq_batched.hpp.
#define MCTQ_BATCHED_PART 2
#include "mctq_batched.hpp"
