// main.cpp - entry point of a setup executable (FluidX3D-master/src/main.cpp:362-381 without window and render thread):
// command-line arguments are device ordinals, then main_setup() runs on this thread.  The keys the reference takes from
// its window (graphics.cpp:623-701) arrive here from whoever started the process:
//   standard input, one character per key: 'P' run/pause, '9' export trigger, 'Q' or Esc stop (a closed or empty stdin is fine);
//   signals: SIGUSR1 = '9', SIGUSR2 = 'P', SIGINT / SIGTERM = stop.
// A setup sees them through the same globals as in the reference: running, key_P, key_9.
#include <csignal>
#include <unistd.h>
#include "setup.hpp"

static void press(const int key) {
	switch (key) {
		case 'P': case 'p': key_P = !key_P; break;
		case '9': key_9 = !key_9; break;
		case 'Q': case 'q': case 27: running = false; break;
		default: break;
	}
}
static void on_signal(const int s) { press(s == SIGUSR1 ? '9' : s == SIGUSR2 ? 'P' : 'Q'); }

int main(int argc, char* argv[]) {
	for (int i = 1; i < argc; i++) main_arguments.push_back(string(argv[i]));
	std::signal(SIGUSR1, on_signal); std::signal(SIGUSR2, on_signal); std::signal(SIGINT, on_signal); std::signal(SIGTERM, on_signal);
	std::thread keys([]() { // ends with the process; read() returns 0 at end of input
		char c;
		while (running && read(STDIN_FILENO, &c, 1) == 1) press((unsigned char)c);
	});
	keys.detach();
	main_setup();
	running = false;
	return 0;
}
