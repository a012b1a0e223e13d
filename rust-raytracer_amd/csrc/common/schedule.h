// The per-tile job sequence of one launch of the path-tracing kernels 1, 2, 5 (host logic; the kernels read it as RenderK::lvl).
#pragma once
#include <cstdint>

namespace rtamd {

static const int SCHED_LEVELS = 4;

// Every tile's samples [s_begin, s_end) are cut into the same sequence of units, in up to SCHED_LEVELS levels of decreasing unit / job
// size.  Level l: rounds (jobs per tile) from lvl[l][0], units from lvl[l][1], samples from lvl[l][2], lvl[l][3] samples per unit,
// lvl[l][4] units per job; the row behind the last level = {rounds, units, s_end, 0, 0} and is repeated up to row SCHED_LEVELS.
struct Schedule {
    int lvl[SCHED_LEVELS + 1][5];
    int units_per_tile;
};

// Fills `s`; returns the number of rounds (jobs per tile).  n_waves: resident waves of the launch.
int make_schedule(Schedule& s, int tiles_owned, int n_waves, int s_begin, int s_end, int sub_spp, int job_units);

}  // namespace rtamd
