// eigen_min.h -- TEST INFRASTRUCTURE (oracle/_ref build only).  The few Eigen types the reference's hot path touches: fixed-size column
// vectors, Map, dot().  Eigen's dot() of a fixed 3-vector is cwiseProduct().sum() through redux_novec_unroller<.., 0, 3>, which splits the
// range in halves: e0 + (e1 + e2) (Eigen/src/Core/Redux.h); restated here for any fixed size by the same recursion.
#pragma once
#include <cstddef>
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW_IF_VECTORIZABLE_FIXED_SIZE(T, N)
namespace Eigen {
enum { ColMajor = 0, RowMajor = 1, Default = 0, Dynamic = -1 };
template<typename T, int R, int C, int O = 0> class Matrix;
template<typename M> class Map;
namespace internal {
template<typename T, int Start, int Length> struct redux_sum_prod {
	enum { Half = Length / 2 };
	static T run(const T* a, const T* b) { return redux_sum_prod<T, Start, Half>::run(a, b) + redux_sum_prod<T, Start + Half, Length - Half>::run(a, b); }
};
template<typename T, int Start> struct redux_sum_prod<T, Start, 1> { static T run(const T* a, const T* b) { return a[Start] * b[Start]; } };
}
template<typename T, int R, int C, int O> class Matrix {
public:
	typedef T Scalar;
	enum { SizeAtCompileTime = R * C };
	Matrix() {}
	Matrix(const T& x, const T& y, const T& z) { static_assert(R * C == 3, ""); v[0] = x; v[1] = y; v[2] = z; }
	template<typename M> Matrix(const Map<M>& m) { for (int i = 0; i < R * C; ++i) v[i] = m.p[i]; }
	template<typename M> Matrix& operator = (const Map<M>& m) { for (int i = 0; i < R * C; ++i) v[i] = m.p[i]; return *this; }
	T dot(const Matrix& o) const { return internal::redux_sum_prod<T, 0, R * C>::run(v, o.v); }
	const T* data() const { return v; }
	T* data() { return v; }
	const T& operator [](size_t i) const { return v[i]; }
	T& operator [](size_t i) { return v[i]; }
	const T& operator ()(size_t i) const { return v[i]; }
	T& operator ()(size_t i) { return v[i]; }
	T v[R * C];
};
template<typename M> class Map {
public:
	typedef typename M::Scalar Scalar;
	explicit Map(Scalar* q) : p(q) {}
	Map& operator = (const M& m) { for (int i = 0; i < (int)M::SizeAtCompileTime; ++i) p[i] = m.v[i]; return *this; }
	operator M () const { M m; for (int i = 0; i < (int)M::SizeAtCompileTime; ++i) m.v[i] = p[i]; return m; }
	Scalar* p;
};
template<typename M> class Map<const M> {
public:
	typedef typename M::Scalar Scalar;
	explicit Map(const Scalar* q) : p(q) {}
	operator M () const { M m; for (int i = 0; i < (int)M::SizeAtCompileTime; ++i) m.v[i] = p[i]; return m; }
	const Scalar* p;
};
}
