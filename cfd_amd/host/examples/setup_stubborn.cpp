// A setup whose loop never looks at `running`: after the stop request LBM::run returns at once without advancing, so this
// loop would spin forever; the process must still end on SIGINT / SIGTERM (main.cpp's watchdog).
#include "../setup.hpp"
void main_setup() {
	LBM lbm(32u, 32u, 32u, 0.01f);
	println("started");
	while (lbm.get_t() < 1000000000000ull) lbm.run(10u);
}
