// klib_sort.hpp -- klib's introsort (cstl/ksort.h:146-226, ks_introsort) restated as a template.  The reference sorts chains by weight and
// alignment regions by end position / score with it; the sort is not stable, the keys tie often, and what happens next depends on the
// order among equal keys -- so the same steps are taken here: median of first / middle+1 / last as pivot, parked at the end; partitions of
// 16 or fewer elements left alone and finished by ONE insertion sort over the whole array; comb sort (shrink factor 1.2473..., gaps 9
// and 10 replaced by 11) for a partition reached after 2 log2(n) levels.  `lt(a, b)`: a sorts before b.
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

template <class T, class Lt> void cs_klib_insertion_(T *s, T *t, Lt lt)
{
	for (T *i = s + 1; i < t; ++i)
		for (T *j = i; j > s && lt(*j, *(j - 1)); --j) std::swap(*j, *(j - 1));
}
template <class T, class Lt> void cs_klib_combsort_(size_t n, T *a, Lt lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	size_t gap = n; bool swapped;
	do {
		if (gap > 2) { gap = (size_t)((double)gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = false;
		for (T *i = a; i < a + n - gap; ++i) if (lt(i[gap], *i)) { std::swap(*i, i[gap]); swapped = true; }
	} while (swapped || gap > 2);
	if (gap != 1) cs_klib_insertion_(a, a + n, lt);
}
template <class T, class Lt> void cs_klib_introsort(size_t n, T *a, Lt lt)
{
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) std::swap(a[0], a[1]); return; }
	int d = 2;
	while ((1ul << d) < n) ++d;
	struct Frame { T *lo, *hi; int depth; };
	std::vector<Frame> stack; stack.reserve(sizeof(size_t) * (size_t)d + 2);
	T *s = a, *t = a + (n - 1);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { cs_klib_combsort_((size_t)(t - s) + 1, s, lt); t = s; continue; }
			T *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
			if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
			else k = lt(*j, *i) ? i : j;
			const T pivot = *k;
			if (k != t) std::swap(*k, *t);
			for (;;) {
				do ++i; while (lt(*i, pivot));
				do --j; while (i <= j && lt(pivot, *j));
				if (j <= i) break;
				std::swap(*i, *j);
			}
			std::swap(*i, *t);
			if (i - s > t - i) {
				if (i - s > 16) stack.push_back({s, i - 1, d});
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) stack.push_back({i + 1, t, d});
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (stack.empty()) { cs_klib_insertion_(a, a + n, lt); return; }
			s = stack.back().lo; t = stack.back().hi; d = stack.back().depth; stack.pop_back();
		}
	}
}
