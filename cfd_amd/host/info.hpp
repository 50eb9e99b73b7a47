// info.hpp - run-time metrics of the host (FluidX3D-master/src/info.hpp, info.cpp:16-21, 105-114): smoothed step time
// and the MLUPs/s definition.  The console table of the reference is not reproduced.
#pragma once
#include <mutex>
#include "utilities.hpp"

class LBM;
struct Info {
	double runtime_lbm = 0.0;                   // accumulated wall time of do_time_step()
	double runtime_lbm_timestep_last = 1.0;     // last step
	double runtime_lbm_timestep_smooth = 1.0;   // exponentially smoothed step time
	ulong steps = max_ulong, steps_last = 0ull;
	std::mutex allow_printing;
	Clock since_start;                          // wall clock since the program started (runtime_total of the reference)
	double runtime_total() { return since_start.stop(); }
	void append(const ulong steps, const ulong total_steps, const ulong t) { // info.cpp:6-15: total_steps, when given, is the length of the whole run
		if (total_steps == max_ulong) { this->steps = steps; steps_last = t; } else this->steps = total_steps;
	}
	void update(const double dt) { // dt_smooth = (dt+0.3)/(0.3/dt_smooth+1)
		runtime_lbm_timestep_last = dt;
		runtime_lbm_timestep_smooth = (dt+0.3)/(0.3/runtime_lbm_timestep_smooth+1.0);
		runtime_lbm += dt;
	}
	void print_initialize(LBM*) {}
	void print_update() {}
	void print_finalize() {}
};
extern Info info; // defined in lbm.cpp
