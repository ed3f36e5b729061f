// sitrk_sort.hip -- device key/value radix sort used to keep buoys ordered by
// host cell (a wavefront's 64 buoys then read a handful of contiguous cell
// records).  Kept in its own translation unit: the rocPRIM template
// instantiation is the slow part of the build.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "sitrk_internal.h"

namespace sitrk {

// two-call protocol: tmp == nullptr -> only *tmp_bytes is written
hipError_t sort_pairs_u32(void *tmp, size_t *tmp_bytes, const uint32_t *kin, uint32_t *kout,
                          const int32_t *vin, int32_t *vout, size_t n, unsigned end_bit, hipStream_t s)
{
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, 0u, end_bit, s);
}

}  // namespace sitrk
