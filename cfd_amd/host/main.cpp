// main.cpp - entry point of a setup executable (FluidX3D-master/src/main.cpp:362-381 without window and render thread):
// command-line arguments are device ordinals, then main_setup() runs on this thread.
#include "setup.hpp"
int main(int argc, char* argv[]) {
	for (int i = 1; i < argc; i++) main_arguments.push_back(string(argv[i]));
	main_setup();
	running = false;
	return 0;
}
