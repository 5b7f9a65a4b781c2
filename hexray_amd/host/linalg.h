// Host-side FP64 vector / matrix helpers for the scene flattener.
// Semantics follow the reference's Vector / Matrix / Transform (reference src/vector.h:30-165,
// src/matrix.h:29-99, src/matrix.cpp:29-187): row vectors, v' = v * M; Transform composes
// scale / rotate on the right of m and recomputes the inverse each time.
#pragma once
#include <cmath>
#include "hexray_hip.h"

namespace hxh {

constexpr double kPi = 3.141592653589793238;   // reference src/constants.h:28
constexpr double kInf = 1e99;                   // reference src/constants.h:29

struct V3 {
	double x = 0, y = 0, z = 0;
	V3() = default;
	V3(double a, double b, double c) : x(a), y(b), z(c) {}
	double& operator[](int i) { return (&x)[i]; }
	const double& operator[](int i) const { return (&x)[i]; }
};

inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(const V3& a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(const V3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, const V3& a) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& a, const V3& b)
{
	return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(const V3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline double length_sqr(const V3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// Vector::normalize(): multiply by the reciprocal of the length (src/vector.h:79-83)
inline V3 unit(const V3& a) { double k = 1.0 / length(a); return {a.x * k, a.y * k, a.z * k}; }
// free normalize(): leaves near-unit vectors untouched (src/vector.h:160-165)
inline V3 unit_lazy(const V3& a)
{
	double len = length(a);
	if (std::fabs(len - 1.0) < 1e-6) return a;
	return a * (1 / len);
}
inline double dist(const V3& a, const V3& b)
{
	return std::sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z));
}

struct M3 {
	double m[3][3];
	static M3 diag(double d)
	{
		M3 r;
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) r.m[i][j] = (i == j) ? d : 0.0;
		return r;
	}
};

inline V3 operator*(const V3& v, const M3& a)
{
	return {v.x * a.m[0][0] + v.y * a.m[1][0] + v.z * a.m[2][0],
	        v.x * a.m[0][1] + v.y * a.m[1][1] + v.z * a.m[2][1],
	        v.x * a.m[0][2] + v.y * a.m[1][2] + v.z * a.m[2][2]};
}

inline M3 operator*(const M3& a, const M3& b)
{
	M3 c = M3::diag(0.0);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++)
			for (int k = 0; k < 3; k++) c.m[i][j] += a.m[i][k] * b.m[k][j];
	return c;
}

inline double det(const M3& a)
{
	return a.m[0][0] * a.m[1][1] * a.m[2][2] - a.m[0][0] * a.m[1][2] * a.m[2][1] -
	       a.m[0][1] * a.m[1][0] * a.m[2][2] + a.m[0][1] * a.m[1][2] * a.m[2][0] +
	       a.m[0][2] * a.m[1][0] * a.m[2][1] - a.m[0][2] * a.m[1][1] * a.m[2][0];
}

// adjugate / determinant, with the reference's "not invertible -> return input" escape (src/matrix.cpp:107-117)
inline M3 inverse(const M3& a)
{
	double D = det(a);
	if (std::fabs(D) < 1e-12) return a;
	double rD = 1.0 / D;
	M3 r;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			// cofactor of (j, i)
			int rr[2], cc[2], nr = 0, nc = 0;
			for (int k = 0; k < 3; k++) {
				if (k != j) rr[nr++] = k;
				if (k != i) cc[nc++] = k;
			}
			double t = a.m[rr[0]][cc[0]] * a.m[rr[1]][cc[1]] - a.m[rr[1]][cc[0]] * a.m[rr[0]][cc[1]];
			if ((i + j) % 2) t = -t;
			r.m[i][j] = rD * t;
		}
	return r;
}

inline M3 transposed(const M3& a)
{
	M3 r;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
	return r;
}

inline double radians(double deg) { return deg / 180.0 * kPi; }

inline M3 rot_x(double ang)
{
	double S = std::sin(ang), C = std::cos(ang);
	M3 a = M3::diag(1.0);
	a.m[1][1] = C; a.m[2][1] = S; a.m[1][2] = -S; a.m[2][2] = C;
	return a;
}
inline M3 rot_y(double ang)
{
	double S = std::sin(ang), C = std::cos(ang);
	M3 a = M3::diag(1.0);
	a.m[0][0] = C; a.m[2][0] = -S; a.m[0][2] = S; a.m[2][2] = C;
	return a;
}
inline M3 rot_z(double ang)
{
	double S = std::sin(ang), C = std::cos(ang);
	M3 a = M3::diag(1.0);
	a.m[0][0] = C; a.m[1][0] = S; a.m[0][1] = -S; a.m[1][1] = C;
	return a;
}

struct Xform {
	V3 offset;
	M3 m = M3::diag(1.0), inv = M3::diag(1.0), tinv = M3::diag(1.0);

	void refresh() { inv = inverse(m); tinv = transposed(inv); }
	void scale(double x, double y, double z)
	{
		M3 s = M3::diag(0.0);
		s.m[0][0] = x; s.m[1][1] = y; s.m[2][2] = z;
		m = m * s;
		refresh();
	}
	void rotate(double yaw, double pitch, double roll)   // degrees; Z(roll) * X(pitch) * Y(yaw)
	{
		m = m * rot_z(radians(roll)) * rot_x(radians(pitch)) * rot_y(radians(yaw));
		refresh();
	}
	void translate(const V3& t) { offset = offset + t; }
	V3 point(const V3& p) const { return p * m + offset; }
	V3 unpoint(const V3& p) const { return (p - offset) * inv; }
	V3 dir(const V3& d) const { return unit_lazy(d * m); }

	void store(hxr_transform& o) const
	{
		for (int i = 0; i < 3; i++) o.offset[i] = offset[i];
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				o.m[i * 3 + j] = m.m[i][j];
				o.inv_m[i * 3 + j] = inv.m[i][j];
				o.tinv_m[i * 3 + j] = tinv.m[i][j];
			}
	}
};

inline void store3(double* o, const V3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

}  // namespace hxh
