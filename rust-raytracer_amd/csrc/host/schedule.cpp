// make_schedule: how one launch's samples are cut into units and jobs (see common/schedule.h).
//
// The reference hands whole row bands to worker threads (camera.rs:79-94); here a wave takes JOBS -- consecutive units of one 8x8
// tile -- from a global counter, round by round (all tiles' round k before any tile's round k+1), and the units of a tile are folded
// into the pixel sums in sample order (camera.rs:96-101) through the tile's ticket.  Jobs are dealt round by round, so the sizes at
// the END of the sequence are the sizes of the jobs still running when the queue runs dry, and what a wave has left to do then is
// idle time for every wave that finished before it.  Measured with the uniform schedule (jobs of 2 units x 8 spp; tools-only build
// -DRT_TAIL_STATS): mean idle wave-time at the end of the launch 2.4 ms of 507.9 (whole headline frame), 3.7 ms of 70.9 (one rank's
// eighth of it).  Small units everywhere are no answer (8 -> 4 -> 2 -> 1 spp per unit: 509.6, 518.1, 545.9, 628.1 ms for the frame),
// so only the end of the sequence is tapered: `r` rounds each of single units of sub_spp, sub_spp / 2 and sub_spp / 4 samples, where
// a round of every level lasts long enough to cover the stragglers of the level before it -- r grows as the rank's share of tiles
// shrinks (r = ceil(TAPER_R x waves / tiles)), capped at a quarter of the launch's samples.
#include "../common/schedule.h"

#include <algorithm>

namespace rtamd {

#ifndef TAPER_R
#define TAPER_R 6
#endif

int make_schedule(Schedule& sch, int tiles_owned, int n_waves, int s_begin, int s_end, int sub_spp, int job_units) {
    const int n = s_end - s_begin, sub = sub_spp;
    // the taper's levels: single units of sub_spp (only when the main part deals jobs of several units), sub_spp / 2, sub_spp / 4
    int size[SCHED_LEVELS - 1], n_taper = 0, taper_unit = 0;
    if (job_units > 1) size[n_taper++] = sub;
    if (sub / 2 >= 1) size[n_taper++] = sub / 2;
    if (sub / 4 >= 1) size[n_taper++] = sub / 4;
    for (int k = 0; k < n_taper; k++) taper_unit += size[k];
    int r = 0;
    if (TAPER_R > 0 && n_taper > 0)
        r = (int)std::min<int64_t>(((int64_t)TAPER_R * n_waves + tiles_owned - 1) / std::max(1, tiles_owned), n / 4 / taper_unit);
    const int main_spp = n - r * taper_unit;
    int round0 = 0, unit0 = 0, s0 = s_begin, l = 0;
    auto level = [&](int spp, int sz, int ju) {  // `spp` samples in units of `sz`, `ju` units per job
        if (spp <= 0) return;
        const int units = (spp + sz - 1) / sz;
        sch.lvl[l][0] = round0; sch.lvl[l][1] = unit0; sch.lvl[l][2] = s0; sch.lvl[l][3] = sz; sch.lvl[l][4] = ju;
        round0 += (units + ju - 1) / ju; unit0 += units; s0 += spp;
        l++;
    };
    level(main_spp, sub, job_units);
    for (int k = 0; k < n_taper; k++) level(r * size[k], size[k], 1);
    for (int k = l; k <= SCHED_LEVELS; k++) { sch.lvl[k][0] = round0; sch.lvl[k][1] = unit0; sch.lvl[k][2] = s_end; sch.lvl[k][3] = 0; sch.lvl[k][4] = 0; }
    sch.units_per_tile = unit0;
    return round0;
}

}  // namespace rtamd
