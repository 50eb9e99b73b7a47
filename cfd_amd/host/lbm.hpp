// lbm.hpp - C++ host with the class surface of the reference's LBM (FluidX3D-master/src/lbm.hpp:208-611), on top of the
// C ABI of liblbmhip.so.  A main_setup() written for the reference compiles against this header: constructors,
// flags/rho/u/F containers with operator[] and .x/.y/.z, run(), update_fields(), read_from_device(), VTK export,
// coordinates()/index()/position()/center()/size().  All lattice arithmetic happens in the HIP kernels.
#pragma once
#include <atomic>
#include <fstream>
#include "defines.hpp"
#include "utilities.hpp"
#include "units.hpp"
#include "info.hpp"
#include "graphics.hpp"
#include "shapes.hpp"
#include "../../include/lbmhip.h"

uint bytes_per_cell_host();                 // FluidX3D-master/src/lbm.cpp:23-36
uint bytes_per_cell_device();               // lbm.cpp:37-50
uint bandwidth_bytes_per_cell_device();     // lbm.cpp:51-82
uint3 resolution(const float3 box_aspect_ratio, const uint memory); // lbm.cpp:83-102: aspect ratio + MB of VRAM -> grid
string default_filename(const string& path, const string& name, const string& extension, const ulong t);
string default_filename(const string& name, const string& extension, const ulong t);
uint lbm_features();                        // extension bits selected in defines.hpp

// host mirror of one device field of one domain (the part of Memory<T>, opencl.hpp:350-612, that LBM uses)
// Host copy of one device field (opencl.hpp:356-372, 505-526).  The buffer is page-locked when the system grants it (the
// transfers then run at the full rate of the host link), ordinary memory otherwise; LBM_PINNED=0 keeps it ordinary.
template <typename T> class Memory {
	lbmhip_domain* domain = nullptr;
	int field = 0;
	ulong N = 0ull;
	uint d = 1u;
	T* host = nullptr;
	bool pinned = false;
	ulong count() const { return N*(ulong)d; }
	void release() {
		if (host) { if (pinned) lbmhip_host_free(host); else std::free(host); }
		host = nullptr; pinned = false;
	}
public:
	Memory() {}
	Memory(lbmhip_domain* domain, const int field, const ulong N, const uint dimensions, const T value = (T)0) : domain(domain), field(field), N(N), d(dimensions) {
		if (count() == 0ull) return;
		const char* env = std::getenv("LBM_PINNED");
		void* p = nullptr;
		if (!(env && env[0] == '0') && lbmhip_host_alloc(count()*sizeof(T), &p) == 0) { host = static_cast<T*>(p); pinned = true; }
		else host = static_cast<T*>(std::malloc(count()*sizeof(T)));
		if (!host) print_error("Host memory of "+to_string((ulong)(count()*sizeof(T)/1048576ull))+" MB could not be allocated.");
		std::fill(host, host+count(), value);
	}
	Memory(const Memory&) = delete;
	Memory& operator=(const Memory&) = delete;
	Memory(Memory&& m) noexcept { *this = std::move(m); }
	Memory& operator=(Memory&& m) noexcept {
		if (this != &m) { release(); domain = m.domain; field = m.field; N = m.N; d = m.d; host = m.host; pinned = m.pinned; m.host = nullptr; m.pinned = false; m.N = 0ull; }
		return *this;
	}
	~Memory() { release(); }
	T* data() { return host; }
	ulong length() const { return N; }
	uint dimensions() const { return d; }
	bool is_pinned() const { return pinned; }
	void reset(const T value = (T)0) { std::fill(host, host+count(), value); write_to_device(); }
	void read_from_device() { if (count() && lbmhip_download(domain, field, host, 0ull, count()*sizeof(T))) print_error(lbmhip_last_error()); }
	void write_to_device() { if (count() && lbmhip_upload(domain, field, host, 0ull, count()*sizeof(T))) print_error(lbmhip_last_error()); }
};

extern std::atomic<unsigned long> lbm_runs_after_stop; // LBM::run calls made although `running` is false

class LBM_Domain { // one lattice block on one GPU (lbm.hpp:20-204)
	uint Nx = 1u, Ny = 1u, Nz = 1u, Dx = 1u, Dy = 1u, Dz = 1u;
	int Ox = 0, Oy = 0, Oz = 0;
	ulong t = 0ull, t_last_update_fields = max_ulong, t_last_force_field = max_ulong;
	float nu = 1.0f/6.0f, fx = 0.0f, fy = 0.0f, fz = 0.0f;
public:
	lbmhip_domain* handle = nullptr;
	Memory<float> rho, u, F;
	Memory<uchar> flags;
	LBM_Domain(const int device, const uint Nx, const uint Ny, const uint Nz, const uint Dx, const uint Dy, const uint Dz, const int Ox, const int Oy, const int Oz, const float nu, const float fx, const float fy, const float fz);
	~LBM_Domain();
	void enqueue_initialize();
	void enqueue_stream_collide();
	void enqueue_stream_collide_planes(const uint z0, const uint z1);
	void enqueue_stream_collide_steps(const uint steps); // `steps` steps with nothing in between (single-domain lattices)
	void enqueue_stream_collide_pair_planes(const uint z0, const uint z1); // step t on [z0, z1) and step t+1 on [z0+2, z1-2)
	void enqueue_update_fields();
	void enqueue_update_force_field();
	void enqueue_update_moving_boundaries();
	void increment_time_step(const uint steps = 1u);
	void reset_time_step();
	void finish_queue();
	uint get_Nx() const { return Nx; }
	uint get_Ny() const { return Ny; }
	uint get_Nz() const { return Nz; }
	ulong get_N() const { return (ulong)Nx*(ulong)Ny*(ulong)Nz; }
	uint get_D() const { return Dx*Dy*Dz; }
	float get_nu() const { return nu; }
	float get_tau() const { return 3.0f*nu+0.5f; }
	float get_fx() const { return fx; }
	float get_fy() const { return fy; }
	float get_fz() const { return fz; }
	ulong get_t() const { return t; }
	uint get_velocity_set() const;
	void set_f(const float fx, const float fy, const float fz) { this->fx = fx; this->fy = fy; this->fz = fz; }
};

class LBM {
	uint Nx = 1u, Ny = 1u, Nz = 1u, Dx = 1u, Dy = 1u, Dz = 1u;
	bool initialized = false;
	vector<lbmhip_domain*> handles;
	void sanity_checks_constructor(const uint D, const float nu, const float fx, const float fy, const float fz, const float sigma, const float alpha, const float beta, const uint particles_N);
	void sanity_checks_initialization();
	void initialize();
	void do_time_step();
	void do_two_time_steps(); // z-slabs: two steps with the inner planes of both in one launch per domain
	void finish_packed_exchange(); // the exchange of the DDF faces do_two_time_steps left packed, on the streams that are current
	void flush_packed_exchange();  // the same on the communication streams, joined
	bool packed_pending = false; ulong packed_t = 0ull;
	bool pairs_slab_steps() const;
	void communicate(const int what);
public:
	// stitches the per-domain host buffers into one global array (Memory_Container, lbm.hpp:239-408)
	template <typename T> class Memory_Container {
		LBM* lbm = nullptr;
		vector<Memory<T>*> buffers;
		string name;
		ulong N = 0ull;
		uint d = 1u;
		uint Nx = 1u, Ny = 1u, Nz = 1u, Dx = 1u, Dy = 1u, Dz = 1u, D = 1u, NxDx = 1u, NyDy = 1u, NzDz = 1u, Hx = 0u, Hy = 0u, Hz = 0u;
		ulong local_Nx = 1ull, local_Ny = 1ull, local_N = 1ull;
		inline T& reference(const ulong i, const uint dimension) { // global index (and component) -> element of the owning domain's buffer
			const ulong n = i%N;
			const uint c = (uint)std::max<ulong>(i/N, (ulong)dimension);
			if (D == 1u) return buffers[0]->data()[n+(ulong)c*N];
			const ulong plane = (ulong)Nx*(ulong)Ny, r = n%plane;
			const uint x = (uint)(r%Nx), y = (uint)(r/Nx), z = (uint)(n/plane);
			const uint domain = x/NxDx+(y/NyDy+z/NzDz*Dy)*Dx;
			const ulong local = (ulong)(x%NxDx+Hx)+((ulong)(y%NyDy+Hy)+(ulong)(z%NzDz+Hz)*local_Ny)*local_Nx;
			return buffers[domain]->data()[local+(ulong)c*local_N];
		}
		string vtk_type() const;
		void write_vtk(const string& path, const bool convert_to_si_units);
	public:
		class Pointer { // lbm.u.x[n], lbm.u.y[n], lbm.u.z[n]
			Memory_Container* memory = nullptr;
			uint dimension = 0u;
		public:
			Pointer() {}
			Pointer(Memory_Container* memory, const uint dimension) : memory(memory), dimension(dimension) {}
			T& operator[](const ulong i) { return memory->reference(i, dimension); }
			const T& operator[](const ulong i) const { return memory->reference(i, dimension); }
		};
		Pointer x, y, z;
		Memory_Container() {}
		Memory_Container(LBM* lbm, const vector<Memory<T>*>& buffers, const string& name);
		Memory_Container& operator=(Memory_Container&& m) noexcept { // re-seat the component pointers on the new object
			lbm = m.lbm; buffers = std::move(m.buffers); name = m.name; N = m.N; d = m.d;
			Nx = m.Nx; Ny = m.Ny; Nz = m.Nz; Dx = m.Dx; Dy = m.Dy; Dz = m.Dz; D = m.D; NxDx = m.NxDx; NyDy = m.NyDy; NzDz = m.NzDz; Hx = m.Hx; Hy = m.Hy; Hz = m.Hz;
			local_Nx = m.local_Nx; local_Ny = m.local_Ny; local_N = m.local_N;
			x = Pointer(this, 0u); y = Pointer(this, 1u); z = Pointer(this, 2u);
			return *this;
		}
		void reset(const T value = (T)0) { for (Memory<T>* b : buffers) b->reset(value); }
		ulong length() const { return N; }
		uint dimensions() const { return d; }
		ulong range() const { return N*(ulong)d; }
		ulong capacity() const { return N*(ulong)d*sizeof(T); }
		T& operator[](const ulong i) { return reference(i, 0u); }
		const T& operator[](const ulong i) const { return const_cast<Memory_Container*>(this)->reference(i, 0u); }
		T operator()(const ulong i) const { return const_cast<Memory_Container*>(this)->reference(i, 0u); }
		T operator()(const ulong i, const uint dimension) const { return const_cast<Memory_Container*>(this)->reference(i, dimension); }
		void read_from_device();  // refreshes rho/u on the device first when UPDATE_FIELDS is off (lbm.hpp:390-396)
		void write_to_device();
		void write_host_to_vtk(const string& path = "", const bool convert_to_si_units = true);
		void write_device_to_vtk(const string& path = "", const bool convert_to_si_units = true);
	};

	vector<LBM_Domain*> lbm_domain; // one per GPU (or several on one GPU when there are fewer GPUs than domains)
	Memory_Container<float> rho;
	Memory_Container<float> u;
	Memory_Container<uchar> flags;
#ifdef FORCE_FIELD
	Memory_Container<float> F;
#endif

	// There is no renderer: camera and frame calls of a setup are accepted and do nothing, so that setups written for the
	// reference's GRAPHICS mode (lbm.hpp:560-611) compile and run headless; next_frame keeps its pacing (lbm.cpp:1900-1914).
	struct Graphics {
		LBM* lbm = nullptr;
		int visualization_modes = 0, field_mode = 0, slice_mode = 0, slice_x = 0, slice_y = 0, slice_z = 0;
		uint last_exported_frame = max_uint;
		void set_camera_centered(const float rx = 0.0f, const float ry = 0.0f, const float fov = 100.0f, const float zoom = 1.0f) {}
		void set_camera_free(const float3& p = float3(0.0f), const float rx = 0.0f, const float ry = 0.0f, const float fov = 100.0f) {}
		bool next_frame(const ulong total_time_steps, const float video_length_seconds) { // true once per frame of a 60 fps video of that length
			const uint frame = to_uint((float)lbm->get_t()/(float)total_time_steps*video_length_seconds*60.0f);
			if (frame == last_exported_frame) return false;
			last_exported_frame = frame;
			return true;
		}
		void print_frame() {}
		void write_frame(const string& path = "", const string& name = "image", const string& extension = ".png", bool print_preview = false) {}
		void write_frame(const uint x1, const uint y1, const uint x2, const uint y2, const string& path = "", const string& name = "image", const string& extension = ".png", bool print_preview = false) {}
		void write_frame_png(const string& path = "", bool print_preview = false) {}
		void write_frame_qoi(const string& path = "", bool print_preview = false) {}
		void write_frame_bmp(const string& path = "", bool print_preview = false) {}
	} graphics;

	LBM(const uint Nx, const uint Ny, const uint Nz, const uint Dx, const uint Dy, const uint Dz, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f, const float sigma = 0.0f, const float alpha = 0.0f, const float beta = 0.0f, const uint particles_N = 0u, const float particles_rho = 0.0f);
	LBM(const uint Nx, const uint Ny, const uint Nz, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f, const float sigma = 0.0f, const float alpha = 0.0f, const float beta = 0.0f, const uint particles_N = 0u, const float particles_rho = 1.0f);
	LBM(const uint3 N, const uint Dx, const uint Dy, const uint Dz, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f, const float sigma = 0.0f, const float alpha = 0.0f, const float beta = 0.0f, const uint particles_N = 0u, const float particles_rho = 0.0f);
	LBM(const uint3 N, const float nu, const float fx = 0.0f, const float fy = 0.0f, const float fz = 0.0f, const float sigma = 0.0f, const float alpha = 0.0f, const float beta = 0.0f, const uint particles_N = 0u, const float particles_rho = 1.0f);
	~LBM();
	LBM(const LBM&) = delete;
	LBM& operator=(const LBM&) = delete;

	void run(const ulong steps = max_ulong, const ulong total_steps = max_ulong);
	void update_fields();
	void reset();
#ifdef FORCE_FIELD
	void update_force_field();
	float3 object_center_of_mass(const uchar flag_marker = TYPE_S);
	float3 object_force(const uchar flag_marker = TYPE_S);
	float3 object_torque(const float3& rotation_center, const uchar flag_marker = TYPE_S);
#endif
#ifdef MOVING_BOUNDARIES
	void update_moving_boundaries();
#endif

	uint get_Nx() const { return Nx; }
	uint get_Ny() const { return Ny; }
	uint get_Nz() const { return Nz; }
	ulong get_N() const { return (ulong)Nx*(ulong)Ny*(ulong)Nz; }
	uint get_Dx() const { return Dx; }
	uint get_Dy() const { return Dy; }
	uint get_Dz() const { return Dz; }
	uint get_D() const { return Dx*Dy*Dz; }
	float get_nu() const { return lbm_domain[0]->get_nu(); }
	float get_tau() const { return 3.0f*get_nu()+0.5f; }
	float get_Re_max() const { return 0.57735027f*std::sqrt((float)(sq(Nx)+sq(Ny)+sq(Nz)))/get_nu(); }
	float get_fx() const { return lbm_domain[0]->get_fx(); }
	float get_fy() const { return lbm_domain[0]->get_fy(); }
	float get_fz() const { return lbm_domain[0]->get_fz(); }
	ulong get_t() const { return lbm_domain[0]->get_t(); }
	uint get_velocity_set() const { return lbm_domain[0]->get_velocity_set(); }
	bool is_initialized() const { return initialized; }
	void set_fx(const float fx) { set_f(fx, get_fy(), get_fz()); }
	void set_fy(const float fy) { set_f(get_fx(), fy, get_fz()); }
	void set_fz(const float fz) { set_f(get_fx(), get_fy(), fz); }
	void set_f(const float fx, const float fy, const float fz) { for (LBM_Domain* d : lbm_domain) d->set_f(fx, fy, fz); }

	void coordinates(const ulong n, uint& x, uint& y, uint& z) const { // n = x+(y+z*Ny)*Nx
		const ulong plane = (ulong)Nx*(ulong)Ny, r = n%plane;
		x = (uint)(r%(ulong)Nx); y = (uint)(r/(ulong)Nx); z = (uint)(n/plane);
	}
	ulong index(const uint x, const uint y, const uint z) const { return (ulong)x+((ulong)y+(ulong)z*(ulong)Ny)*(ulong)Nx; }
	ulong index(const uint3 xyz) const { return index(xyz.x, xyz.y, xyz.z); }
	float3 position(const uint x, const uint y, const uint z) const { return float3((float)x-0.5f*(float)Nx+0.5f, (float)y-0.5f*(float)Ny+0.5f, (float)z-0.5f*(float)Nz+0.5f); }
	float3 position(const ulong n) const { uint x, y, z; coordinates(n, x, y, z); return position(x, y, z); }
	float3 size() const { return float3((float)Nx, (float)Ny, (float)Nz); }
	float3 center() const { return float3(0.5f*(float)Nx-0.5f, 0.5f*(float)Ny-0.5f, 0.5f*(float)Nz-0.5f); }
	uint smallest_side_length() const { return min(min(Nx, Ny), Nz); }
	uint largest_side_length() const { return max(max(Nx, Ny), Nz); }
	float3 relative_position(const uint x, const uint y, const uint z) const { return float3(((float)x+0.5f)/(float)Nx-0.5f, ((float)y+0.5f)/(float)Ny-0.5f, ((float)z+0.5f)/(float)Nz-0.5f); }
	float3 relative_position(const ulong n) const { uint x, y, z; coordinates(n, x, y, z); return relative_position(x, y, z); }
	float3 mirror_position(const float3& p) const { // wraps a position into the periodic box (lbm.hpp:509-515)
		auto wrap = [](const float v, const float n) { return sign(v)*(std::fmod(std::fabs(v)+0.5f*n, n)-0.5f*n); };
		return float3(wrap(p.x, (float)Nx), wrap(p.y, (float)Ny), wrap(p.z, (float)Nz));
	}
	void coordinates(const float3& p, uint& x, uint& y, uint& z) const { // closest lattice point (lbm.hpp:484-489)
		const float3 mp = mirror_position(p);
		x = (uint)(mp.x+1.5f*(float)Nx)%Nx; y = (uint)(mp.y+1.5f*(float)Ny)%Ny; z = (uint)(mp.z+1.5f*(float)Nz)%Nz;
	}
	float get_sigma() const { return 0.0f; } // SURFACE and TEMPERATURE are not provided
	float get_alpha() const { return 0.0f; }
	float get_beta() const { return 0.0f; }
	void write_mesh_to_vtk(const Mesh* mesh, const string& path = "", const bool convert_to_si_units = true) const; // lbm.cpp:1682-1724
	void write_status(const string& path = ""); // lbm.cpp:1599-1643

	void voxelize_mesh_on_device(const Mesh* mesh, const uchar flag = TYPE_S, const float3& rotation_center = float3(0.0f), const float3& linear_velocity = float3(0.0f), const float3& rotational_velocity = float3(0.0f)); // lbm.cpp:1645-1670
	void unvoxelize_mesh_on_device(const Mesh* mesh, const uchar flag = TYPE_S);
	void voxelize_stl(const string& path, const float3& center, const float3x3& rotation, const float size = 0.0f, const uchar flag = TYPE_S); // lbm.cpp:1728-1752
	void voxelize_stl(const string& path, const float3x3& rotation, const float size = 0.0f, const uchar flag = TYPE_S);
	void voxelize_stl(const string& path, const float3& center, const float size = 0.0f, const uchar flag = TYPE_S);
	void voxelize_stl(const string& path, const float size = 0.0f, const uchar flag = TYPE_S);

	friend class LBM_Domain;
};
