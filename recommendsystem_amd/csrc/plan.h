// Layout of the int32 "plan" scratch shared by the sparse-backward and shard-exchange kernels.
#pragma once
#include <stddef.h>

//   sorted_src [F,cap] | entry_chunk [F,cap] | entry_seg [F,cap] | chunk_start [F,cap+1] | seg_row [F,cap] |
//   seg_first_chunk [F,cap+1] | counts [F,4] (n, n_chunk, n_seg, n_padding)
struct PlanView {
    int* sorted_src; int* entry_chunk; int* entry_seg; int* chunk_start; int* seg_row; int* seg_first_chunk;
    int* counts;
};
inline PlanView carve_plan(int* plan, int F, int cap) {
    PlanView v;
    int* p = plan;
    v.sorted_src = p; p += (size_t)F * cap;
    v.entry_chunk = p; p += (size_t)F * cap;
    v.entry_seg = p; p += (size_t)F * cap;
    v.chunk_start = p; p += (size_t)F * (cap + 1);
    v.seg_row = p; p += (size_t)F * cap;
    v.seg_first_chunk = p; p += (size_t)F * (cap + 1);
    v.counts = p;
    return v;
}
