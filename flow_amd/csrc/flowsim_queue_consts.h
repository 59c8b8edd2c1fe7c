// flowsim_queue_consts.h -- host-side constants of the queue-order kernels (flowsim_queue.h), computed by Sim::init_open.
#pragma once

namespace fs {

struct QueueConsts {
  float in_lo[2][2], in_hi[2][2];   // route r: its (at most two) junction-internal stretches [lo, hi); unused: lo = hi = 3e38
  // the segment table of each route (Flow's coordinate of a point = flow + slope * (x - start), O5): at most six
  // segments, unused entries start at 3e38
  float seg_start[2][6], seg_flow[2][6], seg_slope[2][6];
  float veh_len;                    // the one vehicle length
  int ok;                           // the network fits (at most two internal stretches per route, one length)
};

}  // namespace fs
