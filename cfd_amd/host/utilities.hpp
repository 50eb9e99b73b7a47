// utilities.hpp - the small part of the reference's utilities.hpp that setups use (FluidX3D-master/src/utilities.hpp):
// integer typedefs, float3 / float3x3, parallel_for, Clock, console helpers, string helpers.  Written from scratch.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <limits>
#include <string>
#include <thread>
#include <vector>

using std::string;
using std::vector;
using std::thread;
using std::min;
using std::max;
typedef unsigned char uchar;
typedef unsigned short ushort;
typedef unsigned int uint;
typedef uint64_t ulong;
typedef int64_t slong;

#ifndef WORKGROUP_SIZE
#define WORKGROUP_SIZE 64 // some setups size their box in multiples of it (FluidX3D-master/src/opencl.hpp:3)
#endif
constexpr float pif = 3.1415927f;
constexpr double pi = 3.141592653589793;
constexpr uint max_uint = std::numeric_limits<uint>::max();
constexpr ulong max_ulong = std::numeric_limits<ulong>::max();
constexpr double max_double = std::numeric_limits<double>::max();

inline float sq(const float x) { return x*x; }
inline double sq(const double x) { return x*x; }
inline uint sq(const uint x) { return x*x; }
inline float cb(const float x) { return x*x*x; }
inline float radians(const float degrees) { return degrees*(pif/180.0f); }
inline float degrees(const float radians) { return radians*(180.0f/pif); }
inline float sign(const float x) { return x >= 0.0f ? 1.0f : -1.0f; }
inline float rsqrt(const float x) { return 1.0f/std::sqrt(x); }
inline uint to_uint(const float x) { return (uint)std::fmax(std::fmin(x+0.5f, 4294967040.0f), 0.0f); }
inline uint to_uint(const double x) { return (uint)std::fmax(std::fmin(x+0.5, 4294967295.0), 0.0); }
inline ulong to_ulong(const float x) { return (ulong)std::fmax(x+0.5f, 0.0f); }
// a sampled series has settled when both its last difference and its last second difference are below `threshold` times
// the mean of the last three samples (the stop criterion of the reference's validation setups, utilities.hpp:208-213)
template <class T> inline bool converged_impl(const T a, const T b, const T c, const T threshold) {
	const T bound = (a+b+c)/(T)3*threshold;
	return std::fabs(c-b) < bound && std::fabs(c-(T)2*b+a) < bound;
}
inline bool converged(const float x_penultimate, const float x_last, const float x, const float threshold = 1E-5f) { return converged_impl<float>(x_penultimate, x_last, x, threshold); }
inline bool converged(const double x_penultimate, const double x_last, const double x, const double threshold = 1E-5) { return converged_impl<double>(x_penultimate, x_last, x, threshold); }
inline uint gcd(uint a, uint b) { while (b) { const uint t = a%b; a = b; b = t; } return a; }
inline uint lcm(const uint a, const uint b) { return a/gcd(a, b)*b; }

struct uint3 {
	uint x = 0u, y = 0u, z = 0u;
	uint3() {}
	uint3(const uint x, const uint y, const uint z) : x(x), y(y), z(z) {}
	explicit uint3(const uint v) : x(v), y(v), z(v) {}
};
inline uint3 operator/(const uint3& a, const uint b) { return uint3(a.x/b, a.y/b, a.z/b); }
inline uint3 operator*(const uint3& a, const uint b) { return uint3(a.x*b, a.y*b, a.z*b); }

struct float3 {
	float x = 0.0f, y = 0.0f, z = 0.0f;
	float3() {}
	float3(const float x, const float y, const float z) : x(x), y(y), z(z) {}
	explicit float3(const float v) : x(v), y(v), z(v) {}
	float3& operator+=(const float3& b) { x += b.x; y += b.y; z += b.z; return *this; }
	float3& operator-=(const float3& b) { x -= b.x; y -= b.y; z -= b.z; return *this; }
	float3& operator*=(const float s) { x *= s; y *= s; z *= s; return *this; }
	float3& operator/=(const float s) { x /= s; y /= s; z /= s; return *this; }
	float3(const uint3& v) : x((float)v.x), y((float)v.y), z((float)v.z) {}
};
inline float3 operator+(const float3& a, const float3& b) { return float3(a.x+b.x, a.y+b.y, a.z+b.z); }
inline float3 operator-(const float3& a, const float3& b) { return float3(a.x-b.x, a.y-b.y, a.z-b.z); }
inline float3 operator-(const float3& a) { return float3(-a.x, -a.y, -a.z); }
inline float3 operator*(const float s, const float3& a) { return float3(s*a.x, s*a.y, s*a.z); }
inline float3 operator*(const float3& a, const float s) { return float3(s*a.x, s*a.y, s*a.z); }
inline float3 operator/(const float3& a, const float s) { return float3(a.x/s, a.y/s, a.z/s); }
inline float dot(const float3& a, const float3& b) { return a.x*b.x+a.y*b.y+a.z*b.z; }
inline float3 cross(const float3& a, const float3& b) { return float3(a.y*b.z-a.z*b.y, a.z*b.x-a.x*b.z, a.x*b.y-a.y*b.x); }
inline float length(const float3& a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(const float3& a, const float s = 1.0f) { const float l = length(a); return l > 0.0f ? a*(s/l) : a; }

struct float3x3 { // row-major 3x3 matrix; float3x3(axis, angle) is the rotation about `axis` by `angle` radians
	float xx = 1.0f, xy = 0.0f, xz = 0.0f, yx = 0.0f, yy = 1.0f, yz = 0.0f, zx = 0.0f, zy = 0.0f, zz = 1.0f;
	float3x3() {}
	explicit float3x3(const float d) : xx(d), yy(d), zz(d) {}
	float3x3(const float xx, const float xy, const float xz, const float yx, const float yy, const float yz, const float zx, const float zy, const float zz)
		: xx(xx), xy(xy), xz(xz), yx(yx), yy(yy), yz(yz), zx(zx), zy(zy), zz(zz) {}
	float3x3(const float3& axis, const float angle) { // Rodrigues rotation formula
		const float3 n = normalize(axis);
		const float c = std::cos(angle), s = std::sin(angle), t = 1.0f-c;
		xx = t*n.x*n.x+c;     xy = t*n.x*n.y-s*n.z; xz = t*n.x*n.z+s*n.y;
		yx = t*n.y*n.x+s*n.z; yy = t*n.y*n.y+c;     yz = t*n.y*n.z-s*n.x;
		zx = t*n.z*n.x-s*n.y; zy = t*n.z*n.y+s*n.x; zz = t*n.z*n.z+c;
	}
};
inline float3 operator*(const float3x3& m, const float3& v) { return float3(m.xx*v.x+m.xy*v.y+m.xz*v.z, m.yx*v.x+m.yy*v.y+m.yz*v.z, m.zx*v.x+m.zy*v.y+m.zz*v.z); }
inline float3x3 operator*(const float3x3& a, const float3x3& b) {
	return float3x3(a.xx*b.xx+a.xy*b.yx+a.xz*b.zx, a.xx*b.xy+a.xy*b.yy+a.xz*b.zy, a.xx*b.xz+a.xy*b.yz+a.xz*b.zz,
	                a.yx*b.xx+a.yy*b.yx+a.yz*b.zx, a.yx*b.xy+a.yy*b.yy+a.yz*b.zy, a.yx*b.xz+a.yy*b.yz+a.yz*b.zz,
	                a.zx*b.xx+a.zy*b.yx+a.zz*b.zx, a.zx*b.xy+a.zy*b.yy+a.zz*b.zy, a.zx*b.xz+a.zy*b.yz+a.zz*b.zz);
}

// ---- strings and console
inline string to_string(const string& s) { return s; }
inline string to_string(const char* s) { return string(s); }
inline string to_string(const uint x) { return std::to_string(x); }
inline string to_string(const int x) { return std::to_string(x); }
inline string to_string(const ulong x) { return std::to_string((unsigned long long)x); }
inline string to_string(const slong x) { return std::to_string((long long)x); }
// d.dddd[E<exponent>] with the mantissa in [1,10) and the exponent left out when it is 0: the text format of the reference's
// to_string(float) / to_string(double): "d.dddddddd[E<exponent>]" with 8 or 16 decimals.  The reference extracts the digits
// itself, in the arithmetic of the number's own type (FluidX3D-master/src/utilities.hpp:2599-2668, 2745-2765): scale into
// [1, 10) by powers 10^(2^k), cut the leading digit, multiply the rest by 10^decimals, round half up.  Those float products
// are rounded, so the last digits can differ from a correctly rounded conversion (24.87301 prints as 2.48730112E1, printf
// gives 2.48730106E1); VTK headers and status files carry these strings, so the same steps are taken here.
template <typename F> struct decimal_steps;
template <> struct decimal_steps<float> {
	static constexpr int count = 6, decimals = 8;
	static constexpr float up[6] = {1E32f, 1E16f, 1E8f, 1E4f, 1E2f, 1E1f}, down[6] = {1E-32f, 1E-16f, 1E-8f, 1E-4f, 1E-2f, 1E-1f};
	static constexpr float below[6] = {1E-31f, 1E-15f, 1E-7f, 1E-3f, 1E-1f, 1E0f}; // x < 10^-(e-1) is lifted by 10^e
	static constexpr int exponent[6] = {32, 16, 8, 4, 2, 1};
	static constexpr float scale = 1E8f;
	typedef uint digits_t;
};
template <> struct decimal_steps<double> {
	static constexpr int count = 9, decimals = 16;
	static constexpr double up[9] = {1E256, 1E128, 1E64, 1E32, 1E16, 1E8, 1E4, 1E2, 1E1}, down[9] = {1E-256, 1E-128, 1E-64, 1E-32, 1E-16, 1E-8, 1E-4, 1E-2, 1E-1};
	static constexpr double below[9] = {1E-255, 1E-127, 1E-63, 1E-31, 1E-15, 1E-7, 1E-3, 1E-1, 1E0};
	static constexpr int exponent[9] = {256, 128, 64, 32, 16, 8, 4, 2, 1};
	static constexpr double scale = 1E16;
	typedef ulong digits_t;
};
inline string zero_padded(ulong v, const int digits) { string r((size_t)digits, '0'); for (int k = digits-1; k >= 0; k--) { r[(size_t)k] = (char)('0'+v%10ull); v /= 10ull; } return r; }
template <typename F> inline string scientific(F x) {
	typedef decimal_steps<F> S;
	string sign;
	if (x < (F)0) { sign = "-"; x = -x; }
	if (std::isnan(x)) return sign+"NaN";
	if (std::isinf(x)) return sign+"Inf";
	int exponent = 0;
	if (x >= (F)10) for (int k = 0; k < S::count; k++) if (x >= S::up[k]) { x *= S::down[k]; exponent += S::exponent[k]; }
	if (x > (F)0 && x <= (F)1) for (int k = 0; k < S::count; k++) if (x < S::below[k]) { x *= S::up[k]; exponent -= S::exponent[k]; }
	uint leading = (uint)x;
	const F rest = (x-(F)leading)*S::scale;
	typename S::digits_t decimal = (typename S::digits_t)rest;
	if (rest-(F)decimal >= (F)0.5) { // round half up; a carry may run through all digits
		decimal++;
		if ((F)decimal >= S::scale) { decimal = 0; if (++leading >= 10u) { leading = 1u; exponent++; } }
	}
	return sign+std::to_string(leading)+"."+zero_padded((ulong)decimal, S::decimals)+(exponent != 0 ? "E"+std::to_string(exponent) : "");
}
inline string to_string(const float x) { return scientific<float>(x); }
inline string to_string(const double x) { return scientific<double>(x); }
// fixed notation with a given number of decimals (utilities.hpp:2766-2787): half of the last place is added, then the
// digits are cut, all in the number's own type
template <typename F> inline string fixed_decimals(F x, const uint decimals) {
	typedef decimal_steps<F> S;
	string sign;
	if (x < (F)0) { sign = "-"; x = -x; }
	if (std::isnan(x)) return sign+"NaN";
	if (std::isinf(x) || x > (F)max_ulong) return sign+"Inf";
	const int kept = (int)std::min(decimals, (uint)S::decimals);
	const F power = (F)std::pow((F)10, (F)kept);
	x += (F)0.5/power;
	const ulong whole = (ulong)x;
	const typename S::digits_t decimal = (typename S::digits_t)((x-(F)whole)*power);
	return sign+std::to_string(whole)+(decimals == 0u ? "" : "."+zero_padded((ulong)decimal, kept));
}
inline string to_string(const float x, const uint decimals) { return fixed_decimals<float>(x, decimals); }
inline string to_string(const double x, const uint decimals) { return fixed_decimals<double>(x, decimals); }
inline uint length(const string& s) { return (uint)s.length(); }
inline string substring(const string& s, const uint start, const uint n) { return s.substr(start, n); }
inline void print(const string& s = "") { std::cout << s; }
inline void println(const string& s = "") { std::cout << s << std::endl; }
inline void println(const float x) { println(to_string(x)); }
inline void print_info(const string& s) { println("| Info: "+s); }
inline void print_warning(const string& s) { println("| Warning: "+s); }
[[noreturn]] inline void print_error(const string& s) { // the reference prints and exits on every error (utilities.hpp:4159-4166)
	std::cerr << "| Error: " << s << std::endl;
	std::exit(1);
}
// "<y>y<ddd>d<hh>h<mm>m<ss>s": units above the first non-zero one are left out, units below it are zero-padded
// (the text the reference prints, FluidX3D-master/src/utilities.hpp:2876-2888), e.g. "7s", "1m05s", "2h03m07s"
inline string print_time(const double seconds) {
	ulong rest = (ulong)(std::max(seconds, 0.0)+0.5);
	const ulong unit[4] = {31556925ull, 86400ull, 3600ull, 60ull};
	const char tag[4] = {'y', 'd', 'h', 'm'};
	const int width[4] = {1, 3, 2, 2};
	string out;
	bool shown = false;
	for (int k = 0; k < 4; k++) {
		const ulong v = rest/unit[k];
		rest %= unit[k];
		if (!shown && v == 0ull) continue;
		char b[32]; std::snprintf(b, sizeof(b), "%0*llu%c", shown ? width[k] : 1, (unsigned long long)v, tag[k]);
		out += b; shown = true;
	}
	char b[32]; std::snprintf(b, sizeof(b), "%0*llus", shown ? 2 : 1, (unsigned long long)rest);
	return out+b;
}
inline void sleep(const double seconds) { std::this_thread::sleep_for(std::chrono::microseconds((long long)(seconds*1E6))); }
inline void wait() {}
string get_exe_path(); // directory of the running executable with trailing '/', defined in lbm.cpp
void create_folder(const string& path_to_file); // creates the directories leading to a file, defined in lbm.cpp
inline string create_file_extension(const string& filename, const string& extension) {
	const size_t dot = filename.rfind('.'), slash = filename.rfind('/');
	const bool has = dot != string::npos && (slash == string::npos || dot > slash);
	return (has ? filename.substr(0, dot) : filename)+(extension.empty() || extension[0] == '.' ? extension : "."+extension);
}

class Clock {
	std::chrono::high_resolution_clock::time_point t0;
public:
	Clock() { start(); }
	void start() { t0 = std::chrono::high_resolution_clock::now(); }
	double stop() const { return std::chrono::duration<double>(std::chrono::high_resolution_clock::now()-t0).count(); }
};

// ---- parallel_for(N, [threads,] f(n[, t])): static split of [0,N) over host threads (utilities.hpp:60-93)
template <class F> inline void parallel_for(const ulong N, const uint threads, F&& f) {
	const uint T = (uint)std::max<ulong>(1ull, std::min<ulong>(threads ? threads : 1u, N));
	auto body = [&](const uint t) {
		const ulong a = N*(ulong)t/(ulong)T, b = N*(ulong)(t+1u)/(ulong)T;
		for (ulong n = a; n < b; n++) {
			if constexpr (std::is_invocable<F, ulong, uint>::value) f(n, t); else f(n);
		}
	};
	if (T == 1u) { body(0u); return; }
	vector<thread> pool;
	for (uint t = 0u; t < T; t++) pool.emplace_back(body, t);
	for (thread& th : pool) th.join();
}
template <class F> inline void parallel_for(const ulong N, F&& f) { parallel_for(N, (uint)thread::hardware_concurrency(), std::forward<F>(f)); }

// uniform pseudo-random numbers in [0, x] from a caller-held seed
inline float random(uint& seed, const float x = 1.0f) { seed = seed*1664525u+1013904223u; return x*(float)(seed>>8)*(1.0f/16777215.0f); }

// ---- triangle mesh and binary STL reader (FluidX3D-master/src/utilities.hpp:4440-4581)
struct Mesh {
	uint triangle_number = 0u;
	float3 center, pmin, pmax;
	vector<float3> p0, p1, p2; // the three corners of every triangle
	Mesh(const uint triangle_number, const float3& center) : triangle_number(triangle_number), center(center), pmin(center), pmax(center), p0(triangle_number), p1(triangle_number), p2(triangle_number) {}
	void find_bounds() {
		pmin = pmax = p0[0];
		for (uint i = 0u; i < triangle_number; i++) for (const float3& p : {p0[i], p1[i], p2[i]}) {
			pmin = float3(std::fmin(p.x, pmin.x), std::fmin(p.y, pmin.y), std::fmin(p.z, pmin.z));
			pmax = float3(std::fmax(p.x, pmax.x), std::fmax(p.y, pmax.y), std::fmax(p.z, pmax.z));
		}
	}
	// rigid and similarity transforms about `center` (utilities.hpp:4470-4496); the bounds follow the mesh
	void scale(const float factor) {
		for (uint i = 0u; i < triangle_number; i++) { p0[i] = factor*(p0[i]-center)+center; p1[i] = factor*(p1[i]-center)+center; p2[i] = factor*(p2[i]-center)+center; }
		pmin = factor*(pmin-center)+center; pmax = factor*(pmax-center)+center;
	}
	void translate(const float3& shift) {
		for (uint i = 0u; i < triangle_number; i++) { p0[i] += shift; p1[i] += shift; p2[i] += shift; }
		center += shift; pmin += shift; pmax += shift;
	}
	void rotate(const float3x3& rotation) {
		for (uint i = 0u; i < triangle_number; i++) { p0[i] = rotation*(p0[i]-center)+center; p1[i] = rotation*(p1[i]-center)+center; p2[i] = rotation*(p2[i]-center)+center; }
		find_bounds();
	}
	void set_center(const float3& c) { center = c; }
	const float3& get_center() const { return center; }
	float3 get_center_of_mass() const { // centroid of the enclosed volume: signed tetrahedra between the origin and every triangle
		double V = 0.0, cx = 0.0, cy = 0.0, cz = 0.0;
		for (uint i = 0u; i < triangle_number; i++) {
			const double dV = (double)dot(p0[i], cross(p1[i], p2[i]))/6.0;
			const float3 quarter = 0.25f*(p0[i]+p1[i]+p2[i]);
			V += dV; cx += dV*(double)quarter.x; cy += dV*(double)quarter.y; cz += dV*(double)quarter.z;
		}
		return float3((float)(cx/V), (float)(cy/V), (float)(cz/V));
	}
	float get_min_size() const { return std::fmin(std::fmin(pmax.x-pmin.x, pmax.y-pmin.y), pmax.z-pmin.z); }
	float get_max_size() const { return std::fmax(std::fmax(pmax.x-pmin.x, pmax.y-pmin.y), pmax.z-pmin.z); }
	float get_scale_for_box_fit(const float3& box) const { return std::fmin(std::fmin(box.x/(pmax.x-pmin.x), box.y/(pmax.y-pmin.y)), box.z/(pmax.z-pmin.z)); }
	float3 get_bounding_box_size() const { return pmax-pmin; }
	float3 get_bounding_box_center() const { return 0.5f*(pmin+pmax); }
};
// rotate, rescale (size 0: largest that fits the box; >0: longest side = size; <0: factor -size) and, with `reposition`,
// move the bounding-box centre to `center` (otherwise `center` is an offset); utilities.hpp:4530-4581
Mesh* read_stl_raw(const string& path, const bool reposition, const float3& box_size, const float3& center, const float3x3& rotation, const float size); // defined in lbm.cpp
inline Mesh* read_stl(const string& path, const float3& box_size, const float3& center, const float3x3& rotation, const float size) { return read_stl_raw(path, true, box_size, center, rotation, size); }
inline Mesh* read_stl(const string& path, const float3& box_size, const float3& center, const float size) { return read_stl_raw(path, true, box_size, center, float3x3(1.0f), size); }
inline Mesh* read_stl(const string& path, const float scale = 1.0f, const float3x3& rotation = float3x3(1.0f), const float3& offset = float3(0.0f)) { return read_stl_raw(path, false, float3(1.0f), offset, rotation, -std::fabs(scale)); }

template <class T> inline T reverse_bytes(const T value) {
	T r; const char* a = (const char*)&value; char* b = (char*)&r;
	for (size_t i = 0; i < sizeof(T); i++) b[i] = a[sizeof(T)-1u-i];
	return r;
}
