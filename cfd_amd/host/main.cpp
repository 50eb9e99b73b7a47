// main.cpp - entry point of a setup executable (FluidX3D-master/src/main.cpp:362-381 without window and render thread):
// command-line arguments are device ordinals, then main_setup() runs on this thread.  The keys the reference takes from
// its window (graphics.cpp:623-701) arrive here from whoever started the process:
//   standard input, one character per key: 'P' run/pause, '9' export trigger, 'Q' or Esc stop (a closed or empty stdin is fine);
//   signals: SIGUSR1 = '9', SIGUSR2 = 'P', SIGINT / SIGTERM = stop.
// A setup sees them through the same globals as in the reference: running, key_P, key_9.
//
// SIGINT / SIGTERM never leave the process unkillable.  The first one clears `running`, so that LBM::run returns at its
// next batch and a setup that tests `running` can export and leave - however long its exports take; the handler is installed
// with SA_RESETHAND, so a second signal has the default effect (the reference does not intercept these signals at all,
// main.cpp:362-381).  A setup loop such as `while(lbm.get_t()<T) lbm.run(n);` that never looks at `running` would spin forever
// on calls that return at once: a watchdog counts those calls (lbm_runs_after_stop, lbm.cpp) and, once the setup has made a
// hundred of them, ends the process with status 128+signal LBM_STOP_GRACE_MS (default 5000) later.
#include <atomic>
#include <chrono>
#include <csignal>
#include <cstdlib>
#include <unistd.h>
#include "setup.hpp"

static volatile std::sig_atomic_t stop_signal = 0;

static void press(const int key) {
	switch (key) {
		case 'P': case 'p': key_P = !key_P; break;
		case '9': key_9 = !key_9; break;
		case 'Q': case 'q': case 27: running = false; break;
		default: break;
	}
}
static void on_key_signal(const int s) { press(s == SIGUSR1 ? '9' : 'P'); }
static void on_stop_signal(const int s) { stop_signal = s; running = false; }

static void install(const int sig, void (*handler)(int), const int flags) {
	struct sigaction sa;
	sa.sa_handler = handler;
	sigemptyset(&sa.sa_mask);
	sa.sa_flags = flags;
	sigaction(sig, &sa, nullptr);
}

int main(int argc, char* argv[]) {
	for (int i = 1; i < argc; i++) main_arguments.push_back(string(argv[i]));
	install(SIGUSR1, on_key_signal, SA_RESTART); install(SIGUSR2, on_key_signal, SA_RESTART);
	install(SIGINT, on_stop_signal, SA_RESETHAND); install(SIGTERM, on_stop_signal, SA_RESETHAND);
	long grace_ms = 5000;
	if (const char* env = std::getenv("LBM_STOP_GRACE_MS")) grace_ms = std::max(0l, std::atol(env));
	std::thread watchdog([grace_ms]() { // ends with the process
		while (!stop_signal) std::this_thread::sleep_for(std::chrono::milliseconds(50));
		const int s = stop_signal;
		while (lbm_runs_after_stop.load() < 100ul) std::this_thread::sleep_for(std::chrono::milliseconds(50)); // a setup that has stopped stepping is left to finish its exports
		std::this_thread::sleep_for(std::chrono::milliseconds(grace_ms));
		_exit(128+s);
	});
	watchdog.detach();
	std::thread keys([]() { // ends with the process; read() returns 0 at end of input
		char c;
		while (running && read(STDIN_FILENO, &c, 1) == 1) press((unsigned char)c);
	});
	keys.detach();
	main_setup();
	running = false;
	return stop_signal ? 128+stop_signal : 0;
}
