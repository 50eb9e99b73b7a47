// shapes.hpp - geometry predicates for filling the flag lattice: is the lattice point (x, y, z) part of the body?
// Same names, arguments and meaning as FluidX3D-master/src/shapes.hpp:5-17 / shapes.cpp:5-96 (the free-surface fill-level
// variants are not provided).  Bodies are centred on p; axis vectors n carry the body's length in their norm; shells
// (pipe, conepipe, torus) are sqrt(3)/2 cells thick on either side of the nominal radius, which closes them for all
// lattice directions.
#pragma once
#include "utilities.hpp"

namespace shape_detail {
constexpr float shell = 0.8660254f; // sqrt(3)/2
inline float3 rel(const uint x, const uint y, const uint z, const float3& p) { return float3((float)x, (float)y, (float)z)-p; }
// squared distance of t from the axis n through the origin, and the squared coordinate along it
inline float radial2(const float3& t, const float along) { return sq(t.x)+sq(t.y)+sq(t.z)-sq(along); }
inline bool torus(const float3& t, const float axis2, const float r, const float R) { // axis2 = squared distance from the torus axis
	const float s = sq(t.x)+sq(t.y)+sq(t.z)+sq(R), inner = sq(s-sq(r-shell)), outer = sq(s-sq(r+shell)), ring = 4.0f*sq(R)*axis2;
	return ring <= inner && ring >= outer;
}
}

inline bool sphere(const uint x, const uint y, const uint z, const float3& p, const float r) {
	const float3 t = shape_detail::rel(x, y, z, p);
	return sq(t.x)+sq(t.y)+sq(t.z) <= sq(r);
}
inline bool ellipsoid(const uint x, const uint y, const uint z, const float3& p, const float3& r) {
	const float3 t = shape_detail::rel(x, y, z, p);
	return sq(t.x/r.x)+sq(t.y/r.y)+sq(t.z/r.z) <= 1.0f;
}
inline bool cuboid(const uint x, const uint y, const uint z, const float3& p, const float3& l) {
	const float3 t = shape_detail::rel(x, y, z, p);
	return t.x >= -0.5f*l.x && t.x <= 0.5f*l.x && t.y >= -0.5f*l.y && t.y <= 0.5f*l.y && t.z >= -0.5f*l.z && t.z <= 0.5f*l.z;
}
inline bool cube(const uint x, const uint y, const uint z, const float3& p, const float l) { return cuboid(x, y, z, p, float3(l)); }
// solid of revolution about n with radius r1 at the -n end and r2 at the +n end; hollow = only the wall
inline bool revolved(const uint x, const uint y, const uint z, const float3& p, const float3& n, const float r1, const float r2, const bool hollow) {
	const float3 t = shape_detail::rel(x, y, z, p);
	const float along = dot(normalize(n), t), l = length(n);
	const float r = r1 == r2 ? r1 : r1+(r2-r1)*(along+0.5f*l)/l;
	const float d2 = shape_detail::radial2(t, along);
	const bool within = sq(along) <= sq(0.5f*l);
	return within && (hollow ? d2 >= sq(r-shape_detail::shell) && d2 <= sq(r+shape_detail::shell) : d2 <= sq(r));
}
inline bool cylinder(const uint x, const uint y, const uint z, const float3& p, const float3& n, const float r) { return revolved(x, y, z, p, n, r, r, false); }
inline bool cone(const uint x, const uint y, const uint z, const float3& p, const float3& n, const float r1, const float r2) { return revolved(x, y, z, p, n, r1, r2, false); }
inline bool pipe(const uint x, const uint y, const uint z, const float3& p, const float3& n, const float r) { return revolved(x, y, z, p, n, r, r, true); }
inline bool conepipe(const uint x, const uint y, const uint z, const float3& p, const float3& n, const float r1, const float r2) { return revolved(x, y, z, p, n, r1, r2, true); }
// lattice points whose cell is cut by the triangle: foot point inside (barycentric), distance within half the cell's extent along the normal
inline bool triangle(const uint x, const uint y, const uint z, const float3& p0, const float3& p1, const float3& p2) {
	const float3 t = shape_detail::rel(x, y, z, p0), u = p1-p0, v = p2-p0, normal = normalize(cross(u, v));
	const float tn = dot(t, normal);
	const float3 foot = t-tn*normal;
	const float uu = dot(u, u), uv = dot(u, v), vv = dot(v, v), det = uu*vv-uv*uv, fu = dot(foot, u), fv = dot(foot, v);
	const float w0 = (vv*fu-uv*fv)/det, w1 = (uu*fv-uv*fu)/det, w2 = 1.0f-w0-w1;
	const bool inside = w0 >= 0.0f && w0 <= 1.0f && w1 >= 0.0f && w1 <= 1.0f && w2 >= 0.0f && w2 <= 1.0f;
	return inside && 2.0f*std::fabs(tn) <= std::fabs(normal.x)+std::fabs(normal.y)+std::fabs(normal.z);
}
inline bool plane(const uint x, const uint y, const uint z, const float3& p, const float3& n) { return dot(shape_detail::rel(x, y, z, p), n) <= 0.0f; } // the half space behind the plane
inline bool torus_x(const uint x, const uint y, const uint z, const float3& p, const float r, const float R) { const float3 t = shape_detail::rel(x, y, z, p); return shape_detail::torus(t, sq(t.y)+sq(t.z), r, R); }
inline bool torus_y(const uint x, const uint y, const uint z, const float3& p, const float r, const float R) { const float3 t = shape_detail::rel(x, y, z, p); return shape_detail::torus(t, sq(t.x)+sq(t.z), r, R); }
inline bool torus_z(const uint x, const uint y, const uint z, const float3& p, const float r, const float R) { const float3 t = shape_detail::rel(x, y, z, p); return shape_detail::torus(t, sq(t.x)+sq(t.y), r, R); }
