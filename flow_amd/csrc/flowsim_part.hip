// flowsim_part.hip -- one object of libflowsim.so per (precision, lanes per replica): compiled by flow_amd/build.py with
//   -DFS_PART_T=float|double  and  -DFS_PART_SEG=8|16|32|64  (one wave carries 64 / SEG replicas)
//                              or  -DFS_PART_WIDE=2|4         (one workgroup of 2 / 4 waves per replica)
//                              or  -DFS_PART_QUEUE=1          (the queue-order kernels of the open networks, float32)
// so that the step kernels of the pairs compile in parallel and an edit to one kernel family rebuilds few objects.
#include "flowsim_launch.h"

namespace fsim {
#if defined(FS_PART_QUEUE)
template int Sim<FS_PART_T>::launch_queue(int, const float*, size_t, float*, float*, uint8_t*, int);
template int Sim<FS_PART_T>::launch_dropq(int, const float*, size_t, float*, float*, uint8_t*, int);
#elif defined(FS_PART_WIDE)
template int Sim<FS_PART_T>::launch_wide<FS_PART_WIDE>(int, const uint8_t*, const float*, size_t, float*, float*, uint8_t*, int);
#elif defined(FS_PART_SEG)
template int Sim<FS_PART_T>::launch_seg<FS_PART_SEG>(int, const uint8_t*, const float*, size_t, float*, float*, uint8_t*, int);
#if FS_PART_SEG == 16
template int Sim<FS_PART_T>::launch_policy_loop16(const fs_policy*, int, int, const float*, float*, float*, float*, float*,
                                                  uint8_t*);
#endif
#if FS_PART_SEG == 32
template int Sim<FS_PART_T>::launch_policy_row16(const fs_policy*, int, int, const float*, float*, float*, float*, float*,
                                                 uint8_t*);
#endif
#else
#error "flowsim_part.hip: define FS_PART_T and FS_PART_SEG or FS_PART_WIDE"
#endif
}  // namespace fsim
