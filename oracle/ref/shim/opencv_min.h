// opencv_min.h -- TEST INFRASTRUCTURE (oracle/_ref build only).  A minimal restatement of the OpenCV core types the reference's hot path
// is written against (cv::Point_, Point3_, Size_, Matx, Vec and their operators), so that the reference's own sources compile here, where
// OpenCV is not installed.  Arithmetic follows opencv2/core/matx.hpp and types.hpp (4.x): accumulation orders, accumulator types, and the
// overload set that decides which function a call in the reference binds to, are restated from those headers; nothing here is product code.
#pragma once
#include <cmath>
#include <math.h>
#include <cstddef>
#include <cstdint>
#include <limits>
#define CV_MAJOR_VERSION 4
#define CV_ENABLE_UNROLLED 1
namespace cv {

template<typename _Tp> static inline _Tp saturate_cast(float v) { return _Tp(v); }
template<typename _Tp> static inline _Tp saturate_cast(double v) { return _Tp(v); }
template<typename _Tp> static inline _Tp saturate_cast(int v) { return _Tp(v); }
template<> inline int saturate_cast<int>(float v) { return (int)lrintf(v); }     // cvRound
template<> inline int saturate_cast<int>(double v) { return (int)lrint(v); }

enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };
enum { DECOMP_LU = 0 };

template<typename _Tp> class Size_ {
public:
	typedef _Tp value_type;
	Size_() : width(0), height(0) {}
	Size_(_Tp w, _Tp h) : width(w), height(h) {}
	template<typename _Tp2> Size_(const _Tp2& pt, decltype(&_Tp2::x) = nullptr) : width(pt.x), height(pt.y) {}   // Size_(const Point_<_Tp>&), types.hpp
	template<typename _Tp2> operator Size_<_Tp2>() const { return Size_<_Tp2>(saturate_cast<_Tp2>(width), saturate_cast<_Tp2>(height)); }   // types.hpp
	_Tp area() const { return width * height; }
	bool empty() const { return width <= 0 || height <= 0; }
	_Tp width, height;
};
typedef Size_<int> Size2i;
typedef Size2i Size;
template<typename _Tp> static inline bool operator == (const Size_<_Tp>& a, const Size_<_Tp>& b) { return a.width == b.width && a.height == b.height; }
template<typename _Tp> static inline bool operator != (const Size_<_Tp>& a, const Size_<_Tp>& b) { return !(a == b); }

template<typename _Tp, int m, int n> class Matx;
template<typename _Tp, int cn> class Vec;

// ---- Point_ (types.hpp) ----
template<typename _Tp> class Point_ {
public:
	typedef _Tp value_type;
	Point_() : x(0), y(0) {}
	Point_(_Tp _x, _Tp _y) : x(_x), y(_y) {}
	Point_(const Size_<_Tp>& sz) : x(sz.width), y(sz.height) {}
	template<typename _Tp2> operator Point_<_Tp2>() const { return Point_<_Tp2>(saturate_cast<_Tp2>(x), saturate_cast<_Tp2>(y)); }
	_Tp dot(const Point_& pt) const { return saturate_cast<_Tp>(x*pt.x + y*pt.y); }
	double ddot(const Point_& pt) const { return (double)x*(double)pt.x + (double)y*(double)pt.y; }
	double cross(const Point_& pt) const { return (double)x*pt.y - (double)y*pt.x; }   // types.hpp: the cross product is formed in double
	_Tp x, y;
};
template<typename _Tp> static inline Point_<_Tp>& operator += (Point_<_Tp>& a, const Point_<_Tp>& b) { a.x += b.x; a.y += b.y; return a; }
template<typename _Tp> static inline Point_<_Tp>& operator -= (Point_<_Tp>& a, const Point_<_Tp>& b) { a.x -= b.x; a.y -= b.y; return a; }
template<typename _Tp> static inline double norm(const Point_<_Tp>& pt) { return std::sqrt((double)pt.x*pt.x + (double)pt.y*pt.y); }
template<typename _Tp> static inline bool operator == (const Point_<_Tp>& a, const Point_<_Tp>& b) { return a.x == b.x && a.y == b.y; }
template<typename _Tp> static inline bool operator != (const Point_<_Tp>& a, const Point_<_Tp>& b) { return a.x != b.x || a.y != b.y; }
template<typename _Tp> static inline Point_<_Tp> operator + (const Point_<_Tp>& a, const Point_<_Tp>& b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x + b.x), saturate_cast<_Tp>(a.y + b.y)); }
template<typename _Tp> static inline Point_<_Tp> operator - (const Point_<_Tp>& a, const Point_<_Tp>& b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x - b.x), saturate_cast<_Tp>(a.y - b.y)); }
// types.hpp: Point_ / scalar -- quotient in the promoted type of the operands, then saturate_cast back
template<typename _Tp> static inline Point_<_Tp> operator / (const Point_<_Tp>& a, int b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x / b), saturate_cast<_Tp>(a.y / b)); }
template<typename _Tp> static inline Point_<_Tp> operator / (const Point_<_Tp>& a, float b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x / b), saturate_cast<_Tp>(a.y / b)); }
template<typename _Tp> static inline Point_<_Tp> operator / (const Point_<_Tp>& a, double b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x / b), saturate_cast<_Tp>(a.y / b)); }
template<typename _Tp> static inline Point_<_Tp> operator - (const Point_<_Tp>& a) { return Point_<_Tp>(saturate_cast<_Tp>(-a.x), saturate_cast<_Tp>(-a.y)); }
template<typename _Tp> static inline Point_<_Tp> operator * (const Point_<_Tp>& a, int b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x*b), saturate_cast<_Tp>(a.y*b)); }
template<typename _Tp> static inline Point_<_Tp> operator * (int a, const Point_<_Tp>& b) { return Point_<_Tp>(saturate_cast<_Tp>(b.x*a), saturate_cast<_Tp>(b.y*a)); }
template<typename _Tp> static inline Point_<_Tp> operator * (const Point_<_Tp>& a, float b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x*b), saturate_cast<_Tp>(a.y*b)); }
template<typename _Tp> static inline Point_<_Tp> operator * (float a, const Point_<_Tp>& b) { return Point_<_Tp>(saturate_cast<_Tp>(b.x*a), saturate_cast<_Tp>(b.y*a)); }
template<typename _Tp> static inline Point_<_Tp> operator * (const Point_<_Tp>& a, double b) { return Point_<_Tp>(saturate_cast<_Tp>(a.x*b), saturate_cast<_Tp>(a.y*b)); }
template<typename _Tp> static inline Point_<_Tp> operator * (double a, const Point_<_Tp>& b) { return Point_<_Tp>(saturate_cast<_Tp>(b.x*a), saturate_cast<_Tp>(b.y*a)); }

// ---- Point3_ (types.hpp) ----
template<typename _Tp> class Point3_ {
public:
	typedef _Tp value_type;
	Point3_() : x(0), y(0), z(0) {}
	Point3_(_Tp _x, _Tp _y, _Tp _z) : x(_x), y(_y), z(_z) {}
	explicit Point3_(const Point_<_Tp>& pt) : x(pt.x), y(pt.y), z(_Tp()) {}
	template<typename _Tp2> operator Point3_<_Tp2>() const { return Point3_<_Tp2>(saturate_cast<_Tp2>(x), saturate_cast<_Tp2>(y), saturate_cast<_Tp2>(z)); }
	_Tp dot(const Point3_& pt) const { return saturate_cast<_Tp>(x*pt.x + y*pt.y + z*pt.z); }
	double ddot(const Point3_& pt) const { return (double)x*pt.x + (double)y*pt.y + (double)z*pt.z; }
	Point3_ cross(const Point3_& pt) const { return Point3_<_Tp>(y*pt.z - z*pt.y, z*pt.x - x*pt.z, x*pt.y - y*pt.x); }
	_Tp x, y, z;
};
template<typename _Tp> static inline Point3_<_Tp>& operator += (Point3_<_Tp>& a, const Point3_<_Tp>& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
template<typename _Tp> static inline Point3_<_Tp>& operator -= (Point3_<_Tp>& a, const Point3_<_Tp>& b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
template<typename _Tp> static inline Point3_<_Tp>& operator *= (Point3_<_Tp>& a, int b) { a.x = saturate_cast<_Tp>(a.x*b); a.y = saturate_cast<_Tp>(a.y*b); a.z = saturate_cast<_Tp>(a.z*b); return a; }
template<typename _Tp> static inline Point3_<_Tp>& operator *= (Point3_<_Tp>& a, float b) { a.x = saturate_cast<_Tp>(a.x*b); a.y = saturate_cast<_Tp>(a.y*b); a.z = saturate_cast<_Tp>(a.z*b); return a; }
template<typename _Tp> static inline Point3_<_Tp>& operator *= (Point3_<_Tp>& a, double b) { a.x = saturate_cast<_Tp>(a.x*b); a.y = saturate_cast<_Tp>(a.y*b); a.z = saturate_cast<_Tp>(a.z*b); return a; }
template<typename _Tp> static inline double norm(const Point3_<_Tp>& pt) { return std::sqrt((double)pt.x*pt.x + (double)pt.y*pt.y + (double)pt.z*pt.z); }
template<typename _Tp> static inline bool operator == (const Point3_<_Tp>& a, const Point3_<_Tp>& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
template<typename _Tp> static inline bool operator != (const Point3_<_Tp>& a, const Point3_<_Tp>& b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
template<typename _Tp> static inline Point3_<_Tp> operator + (const Point3_<_Tp>& a, const Point3_<_Tp>& b) { return Point3_<_Tp>(saturate_cast<_Tp>(a.x + b.x), saturate_cast<_Tp>(a.y + b.y), saturate_cast<_Tp>(a.z + b.z)); }
template<typename _Tp> static inline Point3_<_Tp> operator - (const Point3_<_Tp>& a, const Point3_<_Tp>& b) { return Point3_<_Tp>(saturate_cast<_Tp>(a.x - b.x), saturate_cast<_Tp>(a.y - b.y), saturate_cast<_Tp>(a.z - b.z)); }
template<typename _Tp> static inline Point3_<_Tp> operator - (const Point3_<_Tp>& a) { return Point3_<_Tp>(saturate_cast<_Tp>(-a.x), saturate_cast<_Tp>(-a.y), saturate_cast<_Tp>(-a.z)); }
template<typename _Tp> static inline Point3_<_Tp> operator * (const Point3_<_Tp>& a, int b) { return Point3_<_Tp>(saturate_cast<_Tp>(a.x*b), saturate_cast<_Tp>(a.y*b), saturate_cast<_Tp>(a.z*b)); }
template<typename _Tp> static inline Point3_<_Tp> operator * (int a, const Point3_<_Tp>& b) { return Point3_<_Tp>(saturate_cast<_Tp>(b.x*a), saturate_cast<_Tp>(b.y*a), saturate_cast<_Tp>(b.z*a)); }
template<typename _Tp> static inline Point3_<_Tp> operator * (const Point3_<_Tp>& a, float b) { return Point3_<_Tp>(saturate_cast<_Tp>(a.x*b), saturate_cast<_Tp>(a.y*b), saturate_cast<_Tp>(a.z*b)); }
template<typename _Tp> static inline Point3_<_Tp> operator * (float a, const Point3_<_Tp>& b) { return Point3_<_Tp>(saturate_cast<_Tp>(b.x*a), saturate_cast<_Tp>(b.y*a), saturate_cast<_Tp>(b.z*a)); }
template<typename _Tp> static inline Point3_<_Tp> operator * (const Point3_<_Tp>& a, double b) { return Point3_<_Tp>(saturate_cast<_Tp>(a.x*b), saturate_cast<_Tp>(a.y*b), saturate_cast<_Tp>(a.z*b)); }
template<typename _Tp> static inline Point3_<_Tp> operator * (double a, const Point3_<_Tp>& b) { return Point3_<_Tp>(saturate_cast<_Tp>(b.x*a), saturate_cast<_Tp>(b.y*a), saturate_cast<_Tp>(b.z*a)); }

// ---- Matx (matx.hpp) ----
struct Matx_AddOp {}; struct Matx_SubOp {}; struct Matx_ScaleOp {}; struct Matx_MulOp {}; struct Matx_DivOp {}; struct Matx_MatMulOp {}; struct Matx_TOp {};
class Mat {};   // only named in signatures that are never instantiated here
class MatExpr {};

template<typename _Tp, int m, int n> class Matx {
public:
	enum { rows = m, cols = n, channels = rows*cols, shortdim = (m < n ? m : n) };
	typedef _Tp value_type;
	typedef Matx<_Tp, m, n> mat_type;
	Matx() { for (int i = 0; i < channels; i++) val[i] = _Tp(0); }
	explicit Matx(_Tp v0) { val[0] = v0; for (int i = 1; i < channels; i++) val[i] = _Tp(0); }
	Matx(_Tp v0, _Tp v1) { static_assert(channels >= 2, ""); val[0] = v0; val[1] = v1; for (int i = 2; i < channels; i++) val[i] = _Tp(0); }
	Matx(_Tp v0, _Tp v1, _Tp v2) { static_assert(channels >= 3, ""); val[0] = v0; val[1] = v1; val[2] = v2; for (int i = 3; i < channels; i++) val[i] = _Tp(0); }
	Matx(_Tp v0, _Tp v1, _Tp v2, _Tp v3) { static_assert(channels >= 4, ""); val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; for (int i = 4; i < channels; i++) val[i] = _Tp(0); }
	Matx(_Tp v0, _Tp v1, _Tp v2, _Tp v3, _Tp v4, _Tp v5, _Tp v6, _Tp v7, _Tp v8) {
		static_assert(channels >= 9, ""); val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; val[4] = v4; val[5] = v5; val[6] = v6; val[7] = v7; val[8] = v8;
		for (int i = 9; i < channels; i++) val[i] = _Tp(0); }
	explicit Matx(const _Tp* vals) { for (int i = 0; i < channels; i++) val[i] = vals[i]; }
	static Matx all(_Tp alpha) { Matx M; for (int i = 0; i < m*n; i++) M.val[i] = alpha; return M; }
	static Matx zeros() { return all(0); }
	static Matx ones() { return all(1); }
	static Matx eye() { Matx M; for (int i = 0; i < shortdim; i++) M(i,i) = 1; return M; }
	_Tp dot(const Matx<_Tp, m, n>& M) const { _Tp s = 0; for (int i = 0; i < channels; i++) s += val[i]*M.val[i]; return s; }
	double ddot(const Matx<_Tp, m, n>& M) const { double s = 0; for (int i = 0; i < channels; i++) s += (double)val[i]*M.val[i]; return s; }
	template<typename T2> operator Matx<T2, m, n>() const { Matx<T2, m, n> M; for (int i = 0; i < m*n; i++) M.val[i] = saturate_cast<T2>(val[i]); return M; }
	Matx<_Tp, n, m> t() const { return Matx<_Tp, n, m>(*this, Matx_TOp()); }
	Matx<_Tp, n, m> inv(int method = DECOMP_LU, bool* p_is_ok = NULL) const;
	Matx<_Tp, m, n> mul(const Matx<_Tp, m, n>& a) const { return Matx<_Tp, m, n>(*this, a, Matx_MulOp()); }
	const _Tp& operator ()(int row, int col) const { return val[row*n + col]; }
	_Tp& operator ()(int row, int col) { return val[row*n + col]; }
	const _Tp& operator ()(int i) const { static_assert(m == 1 || n == 1, ""); return val[i]; }
	_Tp& operator ()(int i) { static_assert(m == 1 || n == 1, ""); return val[i]; }
	Matx(const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b, Matx_AddOp) { for (int i = 0; i < channels; i++) val[i] = saturate_cast<_Tp>(a.val[i] + b.val[i]); }
	Matx(const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b, Matx_SubOp) { for (int i = 0; i < channels; i++) val[i] = saturate_cast<_Tp>(a.val[i] - b.val[i]); }
	template<typename _T2> Matx(const Matx<_Tp, m, n>& a, _T2 alpha, Matx_ScaleOp) { for (int i = 0; i < channels; i++) val[i] = saturate_cast<_Tp>(a.val[i] * alpha); }
	Matx(const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b, Matx_MulOp) { for (int i = 0; i < channels; i++) val[i] = saturate_cast<_Tp>(a.val[i] * b.val[i]); }
	Matx(const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b, Matx_DivOp) { for (int i = 0; i < channels; i++) val[i] = saturate_cast<_Tp>(a.val[i] / b.val[i]); }
	template<int l> Matx(const Matx<_Tp, m, l>& a, const Matx<_Tp, l, n>& b, Matx_MatMulOp) {
		for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) { _Tp s = 0; for (int k = 0; k < l; k++) s += a(i, k) * b(k, j); val[i*n + j] = s; } }
	Matx(const Matx<_Tp, n, m>& a, Matx_TOp) { for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) val[i*n + j] = a(j, i); }
	_Tp val[m*n];
};
namespace internal {
template<typename _Tp, int m> struct Matx_DetOp;
template<typename _Tp> struct Matx_DetOp<_Tp, 3> { double operator ()(const Matx<_Tp, 3, 3>& a) const {
	return a(0,0)*(a(1,1)*a(2,2) - a(2,1)*a(1,2)) - a(0,1)*(a(1,0)*a(2,2) - a(2,0)*a(1,2)) + a(0,2)*(a(1,0)*a(2,1) - a(2,0)*a(1,1)); } };
}
template<typename _Tp, int m> static inline double determinant(const Matx<_Tp, m, m>& a) { return cv::internal::Matx_DetOp<_Tp, m>()(a); }
// Matx_FastInvOp<_Tp, 3, 3> (matx.hpp / operations.hpp)
template<typename _Tp, int m, int n> inline Matx<_Tp, n, m> Matx<_Tp, m, n>::inv(int, bool* p_is_ok) const {
	static_assert(m == 3 && n == 3, "only the 3x3 closed form is restated");
	const Matx<_Tp, 3, 3>& a = *this; Matx<_Tp, 3, 3> b;
	_Tp d = (_Tp)determinant(a);
	if (d == 0) { if (p_is_ok) *p_is_ok = false; return Matx<_Tp, 3, 3>::zeros(); }
	d = 1/d;
	b(0,0) = (a(1,1) * a(2,2) - a(1,2) * a(2,1)) * d;
	b(0,1) = (a(0,2) * a(2,1) - a(0,1) * a(2,2)) * d;
	b(0,2) = (a(0,1) * a(1,2) - a(0,2) * a(1,1)) * d;
	b(1,0) = (a(1,2) * a(2,0) - a(1,0) * a(2,2)) * d;
	b(1,1) = (a(0,0) * a(2,2) - a(0,2) * a(2,0)) * d;
	b(1,2) = (a(0,2) * a(1,0) - a(0,0) * a(1,2)) * d;
	b(2,0) = (a(1,0) * a(2,1) - a(1,1) * a(2,0)) * d;
	b(2,1) = (a(0,1) * a(2,0) - a(0,0) * a(2,1)) * d;
	b(2,2) = (a(0,0) * a(1,1) - a(0,1) * a(1,0)) * d;
	if (p_is_ok) *p_is_ok = true;
	return b;
}

template<typename _Tp, int cn> class Vec : public Matx<_Tp, cn, 1> {
public:
	typedef _Tp value_type;
	enum { channels = cn };
	Vec() {}
	Vec(_Tp v0) : Matx<_Tp, cn, 1>(v0) {}
	Vec(_Tp v0, _Tp v1) : Matx<_Tp, cn, 1>(v0, v1) {}
	Vec(_Tp v0, _Tp v1, _Tp v2) : Matx<_Tp, cn, 1>(v0, v1, v2) {}
	Vec(_Tp v0, _Tp v1, _Tp v2, _Tp v3) : Matx<_Tp, cn, 1>(v0, v1, v2, v3) {}
	explicit Vec(const _Tp* values) : Matx<_Tp, cn, 1>(values) {}
	Vec(const Matx<_Tp, cn, 1>& a, const Matx<_Tp, cn, 1>& b, Matx_AddOp op) : Matx<_Tp, cn, 1>(a, b, op) {}
	Vec(const Matx<_Tp, cn, 1>& a, const Matx<_Tp, cn, 1>& b, Matx_SubOp op) : Matx<_Tp, cn, 1>(a, b, op) {}
	template<typename _T2> Vec(const Matx<_Tp, cn, 1>& a, _T2 alpha, Matx_ScaleOp op) : Matx<_Tp, cn, 1>(a, alpha, op) {}
	Vec cross(const Vec& v) const {
		static_assert(cn == 3, "");
		return Vec<_Tp, 3>(this->val[1]*v.val[2] - this->val[2]*v.val[1], this->val[2]*v.val[0] - this->val[0]*v.val[2], this->val[0]*v.val[1] - this->val[1]*v.val[0]); }
	const _Tp& operator [](int i) const { return this->val[i]; }
	_Tp& operator[](int i) { return this->val[i]; }
	const _Tp& operator ()(int i) const { return this->val[i]; }
	_Tp& operator ()(int i) { return this->val[i]; }
};

// normL2Sqr (base.hpp), unrolled by four as OpenCV builds it
template<typename _Tp, typename _AccTp> static inline _AccTp normL2Sqr(const _Tp* a, int n) {
	_AccTp s = 0; int i = 0;
#if CV_ENABLE_UNROLLED
	for (; i <= n - 4; i += 4) { _AccTp v0 = a[i], v1 = a[i+1], v2 = a[i+2], v3 = a[i+3]; s += v0*v0 + v1*v1 + v2*v2 + v3*v3; }
#endif
	for (; i < n; i++) { _AccTp v = a[i]; s += v*v; }
	return s;
}
template<typename _Tp, int m, int n> static inline double trace(const Matx<_Tp, m, n>& a) { _Tp s = 0; for (int i = 0; i < (m < n ? m : n); i++) s += a(i,i); return s; }
template<typename _Tp, int m, int n> static inline double norm(const Matx<_Tp, m, n>& M) { return std::sqrt(normL2Sqr<_Tp, double>(M.val, m*n)); }

// Matx operators (matx.hpp)
template<typename _Tp1, typename _Tp2, int m, int n> static inline Matx<_Tp1, m, n>& operator += (Matx<_Tp1, m, n>& a, const Matx<_Tp2, m, n>& b) { for (int i = 0; i < m*n; i++) a.val[i] = saturate_cast<_Tp1>(a.val[i] + b.val[i]); return a; }
template<typename _Tp1, typename _Tp2, int m, int n> static inline Matx<_Tp1, m, n>& operator -= (Matx<_Tp1, m, n>& a, const Matx<_Tp2, m, n>& b) { for (int i = 0; i < m*n; i++) a.val[i] = saturate_cast<_Tp1>(a.val[i] - b.val[i]); return a; }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator + (const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b) { return Matx<_Tp, m, n>(a, b, Matx_AddOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator - (const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b) { return Matx<_Tp, m, n>(a, b, Matx_SubOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n>& operator *= (Matx<_Tp, m, n>& a, int alpha) { for (int i = 0; i < m*n; i++) a.val[i] = saturate_cast<_Tp>(a.val[i] * alpha); return a; }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n>& operator *= (Matx<_Tp, m, n>& a, float alpha) { for (int i = 0; i < m*n; i++) a.val[i] = saturate_cast<_Tp>(a.val[i] * alpha); return a; }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n>& operator *= (Matx<_Tp, m, n>& a, double alpha) { for (int i = 0; i < m*n; i++) a.val[i] = saturate_cast<_Tp>(a.val[i] * alpha); return a; }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator * (const Matx<_Tp, m, n>& a, int alpha) { return Matx<_Tp, m, n>(a, alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator * (const Matx<_Tp, m, n>& a, float alpha) { return Matx<_Tp, m, n>(a, alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator * (const Matx<_Tp, m, n>& a, double alpha) { return Matx<_Tp, m, n>(a, alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator * (int alpha, const Matx<_Tp, m, n>& a) { return Matx<_Tp, m, n>(a, alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator * (float alpha, const Matx<_Tp, m, n>& a) { return Matx<_Tp, m, n>(a, alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator * (double alpha, const Matx<_Tp, m, n>& a) { return Matx<_Tp, m, n>(a, alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n>& operator /= (Matx<_Tp, m, n>& a, float alpha) { for (int i = 0; i < m*n; i++) a.val[i] = a.val[i] / alpha; return a; }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n>& operator /= (Matx<_Tp, m, n>& a, double alpha) { for (int i = 0; i < m*n; i++) a.val[i] = a.val[i] / alpha; return a; }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator / (const Matx<_Tp, m, n>& a, float alpha) { return Matx<_Tp, m, n>(a, 1.f/alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator / (const Matx<_Tp, m, n>& a, double alpha) { return Matx<_Tp, m, n>(a, 1./alpha, Matx_ScaleOp()); }
template<typename _Tp, int m, int n> static inline Matx<_Tp, m, n> operator - (const Matx<_Tp, m, n>& a) { return Matx<_Tp, m, n>(a, -1, Matx_ScaleOp()); }
template<typename _Tp, int m, int n, int l> static inline Matx<_Tp, m, n> operator * (const Matx<_Tp, m, l>& a, const Matx<_Tp, l, n>& b) { return Matx<_Tp, m, n>(a, b, Matx_MatMulOp()); }
template<typename _Tp, int m, int n> static inline Vec<_Tp, m> operator * (const Matx<_Tp, m, n>& a, const Vec<_Tp, n>& b) { Matx<_Tp, m, 1> c(a, b, Matx_MatMulOp()); return (const Vec<_Tp, m>&)(c); }
template<typename _Tp, int m, int n> static inline bool operator == (const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b) { for (int i = 0; i < m*n; i++) if (a.val[i] != b.val[i]) return false; return true; }
template<typename _Tp, int m, int n> static inline bool operator != (const Matx<_Tp, m, n>& a, const Matx<_Tp, m, n>& b) { return !(a == b); }
// Matx x Point (types.hpp)
template<typename _Tp> static inline Point3_<_Tp> operator * (const Matx<_Tp, 3, 3>& a, const Point3_<_Tp>& b) {
	Matx<_Tp, 3, 1> tmp = a * Vec<_Tp,3>((_Tp)b.x, (_Tp)b.y, (_Tp)b.z);
	return Point3_<_Tp>(tmp.val[0], tmp.val[1], tmp.val[2]); }
template<typename _Tp> static inline Point3_<_Tp> operator * (const Matx<_Tp, 3, 3>& a, const Point_<_Tp>& b) {
	Matx<_Tp, 3, 1> tmp = a * Vec<_Tp,3>(b.x, b.y, 1);
	return Point3_<_Tp>(tmp.val[0], tmp.val[1], tmp.val[2]); }
// Vec operators
template<typename _Tp, int cn> static inline Vec<_Tp, cn> operator + (const Vec<_Tp, cn>& a, const Vec<_Tp, cn>& b) { return Vec<_Tp, cn>(a, b, Matx_AddOp()); }
template<typename _Tp, int cn> static inline Vec<_Tp, cn> operator - (const Vec<_Tp, cn>& a, const Vec<_Tp, cn>& b) { return Vec<_Tp, cn>(a, b, Matx_SubOp()); }
template<typename _Tp, int cn> static inline Vec<_Tp, cn> operator * (const Vec<_Tp, cn>& a, float alpha) { return Vec<_Tp, cn>(a, alpha, Matx_ScaleOp()); }
template<typename _Tp, int cn> static inline Vec<_Tp, cn> operator * (const Vec<_Tp, cn>& a, double alpha) { return Vec<_Tp, cn>(a, alpha, Matx_ScaleOp()); }

template<class A, class B> void resize(const A&, B&, Size, double = 0, double = 0, int = INTER_LINEAR);   // declared for signatures only; never instantiated
template <typename M> static inline void swap(M& a, M& b) { M t(a); a = b; b = t; }   // cv::swap(Mat&, Mat&): exchanges the headers
} // namespace cv
