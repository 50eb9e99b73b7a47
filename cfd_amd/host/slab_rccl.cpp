// slab_rccl.cpp - one process per GPU without Python: z-slab decomposition of a periodic D3Q19 lattice, halo exchange
// through lbmhip_halo_exchange_rccl on an RCCL communicator, overlapped with the interior collide.  The C++ counterpart of
// LBM::do_time_step + communicate_fi (FluidX3D-master/src/lbm.cpp:1411-1459, 2120-2186) for ranks in separate processes.
//
//   RANK / WORLD_SIZE / LOCAL_RANK   as set by any launcher (defaults 0 / 1 / RANK)
//   LBM_SLAB_ID_FILE                 file through which rank 0 hands the RCCL unique id to the other ranks (WORLD_SIZE > 1)
//   LBM_SLAB_N, LBM_SLAB_STEPS       cells per axis and GPU (default 256), timed steps (default 200)
//   LBM_SLAB_STORAGE                 FP32 (default), FP16S or FP16C
//   LBM_SLAB_SELF=1                  a single rank plays slab 0 of 2 and is its own neighbour on both sides, which makes its
//                                    slab a periodic lattice; the result is compared bit by bit with a plain periodic domain
#include "../../include/lbmhip.h"
#include <rccl/rccl.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static void check(const int status) { if (status) { std::fprintf(stderr, "| Error: %s\n", lbmhip_last_error()); std::exit(1); } }
static void check_rccl(const ncclResult_t r) { if (r != ncclSuccess) { std::fprintf(stderr, "| Error: RCCL: %s\n", ncclGetErrorString(r)); std::exit(1); } }
static int env_int(const char* name, const int fallback) { const char* v = std::getenv(name); return v ? std::atoi(v) : fallback; }

struct Slab {
	lbmhip_domain* dom = nullptr;
	ncclComm_t comm = nullptr;
	uint32_t Nx, Ny, Lz; // local size, Lz includes the two halo planes when split
	int minus, plus;     // neighbour ranks
	bool split;
	uint64_t t = 0;
	void communicate(const int what) { // extract -> exchange -> insert (lbm.cpp:2162-2181)
		if (!split) return;
		check(lbmhip_halo_pack(dom, what, 2, t));
		check(lbmhip_halo_exchange_rccl(dom, comm, what, 2, minus, plus));
		check(lbmhip_halo_unpack(dom, what, 2, t));
	}
	void step() {
		if (!split) { check(lbmhip_stream_collide(dom, t, 0.0f, 0.0f, 0.0f)); t++; return; }
		check(lbmhip_comm_begin(dom));                                                // on the second stream, after the previous step:
		check(lbmhip_stream_collide_planes(dom, t, 0.0f, 0.0f, 0.0f, 1u, 2u));       // the two planes whose DDFs leave the slab,
		check(lbmhip_stream_collide_planes(dom, t, 0.0f, 0.0f, 0.0f, Lz-2u, Lz-1u));
		communicate(LBMHIP_HALO_FI);                                                  // their exchange ...
		check(lbmhip_comm_end(dom));
		check(lbmhip_stream_collide_planes(dom, t, 0.0f, 0.0f, 0.0f, 2u, Lz-2u));   // ... while the interior collides
		check(lbmhip_comm_join(dom));
		t++;
	}
	// Steps t and t+1 with the inner planes of both in one launch (lbmhip_stream_collide_pair_planes).  Both exchanges run beside an
	// inner launch: that of step t beside this call's, that of step t+1 - packed here - beside the next call's (step t+2 on the planes
	// 2 .. Lz-3 owns no slot that the unpack writes; the planes 1 and Lz-2 follow the unpack on the second stream).
	bool packed = false; uint64_t packed_t = 0;
	void finish_packed() {
		if (!packed) return;
		packed = false;
		check(lbmhip_halo_exchange_rccl(dom, comm, LBMHIP_HALO_FI, 2, minus, plus));
		check(lbmhip_halo_unpack(dom, LBMHIP_HALO_FI, 2, packed_t));
	}
	bool pairs() const { int v = 0; check(lbmhip_get_tuning(dom, "pairs_steps", &v)); return split && Lz >= 12u && v != 0; }
	void two_steps() {
		check(lbmhip_comm_begin(dom));
		finish_packed();
		check(lbmhip_stream_collide_planes(dom, t, 0.0f, 0.0f, 0.0f, 1u, 2u));
		check(lbmhip_stream_collide_planes(dom, t, 0.0f, 0.0f, 0.0f, Lz-2u, Lz-1u));
		communicate(LBMHIP_HALO_FI);
		check(lbmhip_comm_end(dom));
		check(lbmhip_stream_collide_pair_planes(dom, t, 0.0f, 0.0f, 0.0f, 2u, Lz-2u)); // step t on 2 .. Lz-3, step t+1 on 4 .. Lz-5
		check(lbmhip_comm_join(dom));
		t++;
		check(lbmhip_stream_collide_planes(dom, t, 0.0f, 0.0f, 0.0f, 1u, 4u));
		check(lbmhip_stream_collide_planes(dom, t, 0.0f, 0.0f, 0.0f, Lz-4u, Lz-1u));
		check(lbmhip_halo_pack(dom, LBMHIP_HALO_FI, 2, t));
		packed = true; packed_t = t;
		t++;
	}
	void flush_packed() {
		if (!packed) return;
		check(lbmhip_comm_begin(dom));
		finish_packed();
		check(lbmhip_comm_end(dom));
		check(lbmhip_comm_join(dom));
	}
	void run(const int steps) { // pairs of steps where the library runs them in one launch, single steps otherwise
		const bool paired = pairs();
		for (int k = 0; k < steps; ) {
			if (paired && k+1 < steps) { two_steps(); k += 2; } else { flush_packed(); step(); k++; }
		}
		flush_packed();
	}
};

// velocity field periodic over the GLOBAL lattice Nx x Ny x Nzg; plane z of the global lattice
static void fill_plane(std::vector<float>& ux, std::vector<float>& uy, std::vector<float>& uz, const uint32_t Nx, const uint32_t Ny, const uint32_t Nzg, const uint32_t z) {
	const float pi2 = 6.2831853f;
	for (uint32_t y = 0; y < Ny; y++) for (uint32_t x = 0; x < Nx; x++) {
		const float a = pi2*((float)x+0.5f)/(float)Nx, b = pi2*((float)y+0.5f)/(float)Ny, c = pi2*((float)z+0.5f)/(float)Nzg;
		const size_t n = (size_t)y*Nx+x;
		ux[n] = 0.05f*std::cos(a)*std::sin(b)*std::sin(c);
		uy[n] = -0.05f*std::sin(a)*std::cos(b)*std::sin(c);
		uz[n] = 0.02f*std::sin(a)*std::sin(b)*std::cos(c);
	}
}
static void upload_velocity(lbmhip_domain* dom, const uint32_t Nx, const uint32_t Ny, const uint32_t Lz, const uint32_t halo, const uint32_t Nzg, const uint32_t z_origin) {
	const size_t plane = (size_t)Nx*Ny, N = plane*Lz;
	std::vector<float> ux(plane), uy(plane), uz(plane);
	for (uint32_t lz = halo; lz < Lz-halo; lz++) {
		fill_plane(ux, uy, uz, Nx, Ny, Nzg, z_origin+lz-halo);
		check(lbmhip_upload(dom, LBMHIP_FIELD_U, ux.data(), ((size_t)lz*plane)*4u, plane*4u));
		check(lbmhip_upload(dom, LBMHIP_FIELD_U, uy.data(), (N+(size_t)lz*plane)*4u, plane*4u));
		check(lbmhip_upload(dom, LBMHIP_FIELD_U, uz.data(), (2u*N+(size_t)lz*plane)*4u, plane*4u));
	}
}

int main() {
	const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local_rank = env_int("LOCAL_RANK", rank);
	const bool self = env_int("LBM_SLAB_SELF", 0) != 0;
	const uint32_t n = (uint32_t)env_int("LBM_SLAB_N", 256);
	const int steps = env_int("LBM_SLAB_STEPS", 200);
	const char* st = std::getenv("LBM_SLAB_STORAGE");
	const uint32_t storage = st && !std::strcmp(st, "FP16S") ? LBMHIP_FP16S : st && !std::strcmp(st, "FP16C") ? LBMHIP_FP16C : LBMHIP_FP32;
	if (self && world != 1) { std::fprintf(stderr, "| Error: LBM_SLAB_SELF needs a single rank\n"); return 1; }
	int devices = 0;
	check(lbmhip_device_count(&devices));
	if (devices < 1) { std::fprintf(stderr, "| Error: no GPU\n"); return 1; }
	const int device = local_rank%devices;

	Slab s;
	const uint32_t slabs = self ? 2u : (uint32_t)world; // number of z-slabs the global lattice is cut into
	s.split = slabs > 1u;
	s.Nx = n; s.Ny = n; s.Lz = n+(s.split ? 2u : 0u);
	s.minus = self ? 0 : (rank+world-1)%world;
	s.plus = self ? 0 : (rank+1)%world;
	const uint32_t Nzg = self ? n : n*(uint32_t)world; // in self mode the one slab wraps onto itself: the global lattice is the slab
	lbmhip_config cfg;
	std::memset(&cfg, 0, sizeof(cfg));
	cfg.Nx = s.Nx; cfg.Ny = s.Ny; cfg.Nz = s.Lz; cfg.Dx = 1u; cfg.Dy = 1u; cfg.Dz = slabs;
	cfg.Oz = (int32_t)(rank*n)-(s.split ? 1 : 0);
	cfg.velocity_set = 19u; cfg.storage = storage; cfg.collision = LBMHIP_SRT; cfg.features = 0u; cfg.nu = 0.01f; cfg.device = device;
	check(lbmhip_domain_create(&cfg, &s.dom));

	if (s.split) { // RCCL communicator: rank 0 publishes the unique id through a file
		if (hipSetDevice(device) != hipSuccess) { std::fprintf(stderr, "| Error: cannot select device %d\n", device); return 1; }
		ncclUniqueId id;
		if (world > 1) {
			const char* path = std::getenv("LBM_SLAB_ID_FILE");
			if (!path) { std::fprintf(stderr, "| Error: LBM_SLAB_ID_FILE is not set\n"); return 1; }
			if (rank == 0) {
				check_rccl(ncclGetUniqueId(&id));
				const std::string tmp = std::string(path)+".tmp";
				{ std::ofstream f(tmp, std::ios::binary); f.write((const char*)&id, sizeof(id)); }
				std::rename(tmp.c_str(), path);
			} else {
				for (int tries = 0; ; tries++) {
					std::ifstream f(path, std::ios::binary);
					if (f && f.read((char*)&id, sizeof(id))) break;
					if (tries > 600) { std::fprintf(stderr, "| Error: no RCCL id in %s after 60 s\n", path); return 1; }
					std::this_thread::sleep_for(std::chrono::milliseconds(100));
				}
			}
		} else check_rccl(ncclGetUniqueId(&id));
		check_rccl(ncclCommInitRank(&s.comm, world, id, rank));
	}

	// LBM::initialize (lbm.cpp:1354-1409): fields to the device, halo of rho/u/flags, initialize kernel, halos again, DDF halo; t is odd meanwhile
	upload_velocity(s.dom, s.Nx, s.Ny, s.Lz, s.split ? 1u : 0u, Nzg, (uint32_t)rank*n);
	s.t = 1;
	s.communicate(LBMHIP_HALO_RHO_U_FLAGS);
	check(lbmhip_initialize(s.dom));
	s.communicate(LBMHIP_HALO_RHO_U_FLAGS);
	s.communicate(LBMHIP_HALO_FI);
	check(lbmhip_sync(s.dom));
	s.t = 0;

	const int warmup = 100; // covers the library's launch-geometry search
	s.run(warmup);
	check(lbmhip_sync(s.dom));
	const auto t0 = std::chrono::high_resolution_clock::now();
	s.run(steps);
	check(lbmhip_sync(s.dom));
	const double seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now()-t0).count();
	const double mlups = (double)n*n*n*steps/seconds*1E-6;
	std::printf("rank %d of %d on device %d: %ux%ux%u cells, %d steps, %.3f ms/step, %.0f MLUPs/s per GPU%s\n", rank, world, device, n, n, n, steps,
	            seconds/steps*1E3, mlups, !s.split ? "" : s.pairs() ? " (halo exchange over RCCL, overlapped; steps two at a time)" : " (halo exchange over RCCL, overlapped)");

	int result = 0;
	if (self) { // the same lattice as one periodic domain
		lbmhip_config c1 = cfg;
		c1.Nz = n; c1.Dz = 1u; c1.Oz = 0;
		lbmhip_domain* ref = nullptr;
		check(lbmhip_domain_create(&c1, &ref));
		upload_velocity(ref, n, n, n, 0u, n, 0u);
		check(lbmhip_initialize(ref));
		for (int k = 0; k < warmup+steps; k++) check(lbmhip_stream_collide(ref, (uint64_t)k, 0.0f, 0.0f, 0.0f));
		const uint64_t T = (uint64_t)(warmup+steps);
		check(lbmhip_update_fields(ref, T, 0.0f, 0.0f, 0.0f));
		check(lbmhip_update_fields(s.dom, T, 0.0f, 0.0f, 0.0f));
		check(lbmhip_sync(ref)); check(lbmhip_sync(s.dom));
		const size_t plane = (size_t)n*n, N1 = plane*n, N2 = plane*s.Lz;
		std::vector<float> a(N1), b(N1);
		size_t differing = 0;
		double umax = 0.0;
		for (int c = 0; c < 4; c++) { // rho, ux, uy, uz: interior planes of the slab against the periodic domain
			const int field = c == 0 ? LBMHIP_FIELD_RHO : LBMHIP_FIELD_U;
			const size_t comp = c == 0 ? 0u : (size_t)(c-1);
			check(lbmhip_download(ref, field, a.data(), comp*N1*4u, N1*4u));
			check(lbmhip_download(s.dom, field, b.data(), (comp*N2+plane)*4u, N1*4u));
			differing += std::memcmp(a.data(), b.data(), N1*4u) != 0 ? 1u : 0u;
			if (c > 0) for (size_t i = 0; i < N1; i += 97u) umax = std::fmax(umax, std::fabs((double)b[i]));
		}
		std::printf("self-exchange against the periodic domain: %s (max |u| sampled %.4f)\n", differing == 0u && umax > 1E-3 ? "bit-identical" : "DIFFERENT", umax);
		result = differing == 0u && umax > 1E-3 ? 0 : 2;
		check(lbmhip_domain_destroy(ref));
	}
	if (s.comm) check_rccl(ncclCommDestroy(s.comm));
	check(lbmhip_domain_destroy(s.dom));
	return result;
}
