// Test infrastructure (oracle/_ref build, step 1). Prints the OpenCL C text that the reference's
// get_opencl_c_code() (FluidX3D-master/src/kernel.hpp:7-18) returns; linked against the reference's
// kernel.cpp compiled where it lies. The output goes to a temporary directory, never into the repository.
#include <cstdio>
#include <string>
std::string get_opencl_c_code();
int main() {
	const std::string code = get_opencl_c_code();
	std::fwrite(code.data(), 1, code.size(), stdout);
	return 0;
}
