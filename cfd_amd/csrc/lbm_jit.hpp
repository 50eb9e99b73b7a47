// lbm_jit.hpp - run-time compilation of stream_collide for feature sets without a precompiled instance.
//
// The reference builds its OpenCL kernels at start-up with the configured extensions as #defines, so every configuration
// runs code specialised for exactly its feature set (FluidX3D-master/src/lbm.cpp:424-665, opencl.hpp:285-300).  The
// counterpart here: the kernel templates of lbm_kernels.hpp are embedded in the library as text, and the first launch of a
// feature set that has no precompiled instance compiles <Q, storage, collision, features> for the five launch geometries
// with hipRTC (1.5-4 s by velocity set, once per process and device).  Same compiler and flags as the precompiled instances,
// hence the same bits.  hipRTC is looked up at run time; without it, or if the compilation fails, the run-time-feature
// instances are used (same results, 10-50 % slower) and one line on stderr says so.  LBMHIP_JIT=0 switches this off.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace lbm { namespace jit {

extern const char* const kDeviceHeader;  // text of lbm_device.hpp (generated at build time, kernel_sources.inc)
extern const char* const kKernelsHeader; // text of lbm_kernels.hpp

struct Instance { hipFunction_t fn[5]; }; // index: geometry_index(V, P)
inline int geometry_index(int V, int P) { return V == 1 ? 0 : V == 2 ? (P == 2 ? 2 : 1) : (P == 2 ? 4 : 3); }

struct Rtc { // the hipRTC entry points used, resolved with dlsym
	typedef void* Program;
	int (*create)(Program*, const char*, const char*, int, const char* const*, const char* const*);
	int (*add_name)(Program, const char*);
	int (*compile)(Program, int, const char* const*);
	int (*lowered_name)(Program, const char*, const char**);
	int (*code_size)(Program, size_t*);
	int (*code)(Program, char*);
	int (*log_size)(Program, size_t*);
	int (*log)(Program, char*);
	int (*destroy)(Program*);
	bool ok;
};
inline const Rtc& rtc() {
	static const Rtc api = []() {
		Rtc a = {};
		void* h = dlopen("libhiprtc.so", RTLD_NOW|RTLD_LOCAL);
		if (!h) h = dlopen("libhiprtc.so.7", RTLD_NOW|RTLD_LOCAL);
		if (!h) return a;
		a.create = (decltype(a.create))dlsym(h, "hiprtcCreateProgram");
		a.add_name = (decltype(a.add_name))dlsym(h, "hiprtcAddNameExpression");
		a.compile = (decltype(a.compile))dlsym(h, "hiprtcCompileProgram");
		a.lowered_name = (decltype(a.lowered_name))dlsym(h, "hiprtcGetLoweredName");
		a.code_size = (decltype(a.code_size))dlsym(h, "hiprtcGetCodeSize");
		a.code = (decltype(a.code))dlsym(h, "hiprtcGetCode");
		a.log_size = (decltype(a.log_size))dlsym(h, "hiprtcGetProgramLogSize");
		a.log = (decltype(a.log))dlsym(h, "hiprtcGetProgramLog");
		a.destroy = (decltype(a.destroy))dlsym(h, "hiprtcDestroyProgram");
		a.ok = a.create && a.add_name && a.compile && a.lowered_name && a.code_size && a.code && a.log_size && a.log && a.destroy;
		return a;
	}();
	return api;
}

// the compiled instance for (device, Q, ST, COLL, FEAT), or nullptr when it cannot be had (said once per feature set on
// stderr; the compiler's log with LBMHIP_JIT_VERBOSE).  small: D2Q9 / D3Q15, which only run one cell per lane.
// Compilations of different feature sets or devices run side by side: the table is locked only to look an entry up or to
// create it, and each entry has its own lock for the seconds its compilation takes.
struct Entry { std::mutex busy; bool done = false; Instance* inst = nullptr; };
inline const Instance* instance(int device, int Q, int ST, int COLL, int FEAT, bool small) {
	static std::mutex table_lock;
	static std::map<std::tuple<int, int, int, int, int>, Entry*> cache; // failed attempts stay cached with inst == nullptr
	{ const char* e = getenv("LBMHIP_JIT"); if (e && e[0] == '0') return nullptr; }
	Entry* entry = nullptr;
	{
		std::lock_guard<std::mutex> lock(table_lock);
		Entry*& slot = cache[std::make_tuple(device, Q, ST, COLL, FEAT)];
		if (!slot) slot = new Entry();
		entry = slot;
	}
	std::lock_guard<std::mutex> lock(entry->busy);
	if (entry->done) return entry->inst;
	Instance* result = nullptr;
	const bool verbose = getenv("LBMHIP_JIT_VERBOSE") != nullptr;
	const Rtc& api = rtc();
	do {
		if (!api.ok) { if (verbose) std::fprintf(stderr, "lbmhip jit: hipRTC not available\n"); break; }
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, device) != hipSuccess) break;
		char name[5][128];
		std::snprintf(name[0], sizeof(name[0]), "lbm::k_collide<%d, %d, %d, %d, false>", Q, ST, COLL, FEAT);
		const int VP[5][2] = {{1, 1}, {2, 1}, {2, 2}, {4, 1}, {4, 2}};
		for (int k = 1; k < 5; k++) std::snprintf(name[k], sizeof(name[k]), "lbm::k_collide_wide<%d, %d, %d, %d, %d, %d>", Q, ST, COLL, FEAT, VP[k][0], VP[k][1]);
		const int count = small ? 1 : 5;
		std::string source = "#include \"lbm_kernels.hpp\"\n";
		for (int k = 0; k < count; k++) source += std::string("template __global__ void ")+name[k]+"(const lbm::Args);\n";
		const char* headers[2] = {kKernelsHeader, kDeviceHeader};
		const char* include_names[2] = {"lbm_kernels.hpp", "lbm_device.hpp"};
		Rtc::Program prog = nullptr;
		if (api.create(&prog, source.c_str(), "lbm_jit.hip", 2, headers, include_names) != 0) break;
		for (int k = 0; k < count; k++) api.add_name(prog, name[k]);
		const char* rocm = getenv("ROCM_PATH");
		const std::string arch = std::string("--offload-arch=")+prop.gcnArchName, inc = std::string("-I")+(rocm ? rocm : "/opt/rocm")+"/include";
		const char* options[] = {arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", inc.c_str()};
		const int status = api.compile(prog, 5, options);
		if (status != 0) {
			if (verbose) { size_t n = 0; api.log_size(prog, &n); std::string log(n+1u, '\0'); if (n) api.log(prog, &log[0]); std::fprintf(stderr, "lbmhip jit: compilation failed\n%s\n", log.c_str()); }
			api.destroy(&prog);
			break;
		}
		size_t bytes = 0;
		std::vector<char> code;
		hipModule_t module = nullptr;
		Instance* inst = new Instance();
		bool good = api.code_size(prog, &bytes) == 0 && bytes > 0u;
		if (good) { code.resize(bytes); good = api.code(prog, code.data()) == 0; }
		if (good) good = hipModuleLoadData(&module, code.data()) == hipSuccess; // onto the current device (the caller made `device` current)
		for (int k = 0; good && k < 5; k++) {
			inst->fn[k] = nullptr;
			if (k >= count) continue;
			const char* lowered = nullptr;
			good = api.lowered_name(prog, name[k], &lowered) == 0 && lowered && hipModuleGetFunction(&inst->fn[k], module, lowered) == hipSuccess;
		}
		api.destroy(&prog);
		if (!good) {
			(void)hipGetLastError();
			if (module) (void)hipModuleUnload(module);
			delete inst;
			if (verbose) std::fprintf(stderr, "lbmhip jit: loading the compiled module failed\n");
			break;
		}
		if (verbose) std::fprintf(stderr, "lbmhip jit: compiled D?Q%d storage %d collision %d features 0x%x for device %d\n", Q, ST, COLL, (unsigned)FEAT, device);
		result = inst; // the module stays loaded for the life of the process
	} while (false);
	if (!result) std::fprintf(stderr, "lbmhip: no kernel could be compiled for D?Q%d storage %d collision %d features 0x%x on device %d; using the kernels that test the features at run time (same results, slower; LBMHIP_JIT_VERBOSE=1 shows why)\n",
	                          Q, ST, COLL, (unsigned)FEAT, device);
	entry->inst = result;
	entry->done = true;
	return result;
}

} } // namespace lbm::jit
