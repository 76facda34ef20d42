// libstdc++'s std::nth_element, restated (g++ 11 of the image; bits/stl_algo.h, bits/stl_heap.h).
//
// The reference calls std::nth_element in two places whose RESULT depends on the library's data movements, not only on
// the order statistic: CirclesEventFrame.cpp:136-147 (the member of median norm among several of EQUAL norm: which one
// ends up at the nth position depends on the input order and on every swap) and EventCalibIni.cpp:78 (the median row
// angle of the keyframe gate, where an angle can be NaN — every comparison with it is false, the standard calls the
// result unspecified, the library's is whatever these loops leave at the nth position).  So the loops are restated one to
// one: __introselect (median of three moved to the front, __unguarded_partition, depth limit 2 lg n, then __heap_select +
// iter_swap), finished by __insertion_sort on at most three elements.  less(x, y) is the caller's comparison on VALUES of
// the array (indices into a key table for the clusters, doubles for the angles); it must be a pure function.
// Checked against the real std::nth_element on the host (ecal_ref_nth_element_f64, tests/test_oracle_policy.py).
#pragma once
#include <stdint.h>

#ifndef ECAL_HD
#ifdef __HIPCC__
#define ECAL_HD __host__ __device__ __forceinline__
#else
#define ECAL_HD inline
#endif
#endif

namespace ecal {

// The array the routines work on: a pointer, or any accessor with operator[] (a reference-like proxy), operator+ (an offset
// view) and value_type — the extraction's tie path keeps the members of a small cluster in the LANES of one vector register
// and runs these loops wave-uniformly on v_readlane / v_writelane (extract_window.hpp, LaneArr).
template <typename A> struct ref_arr_traits { using value_type = typename A::value_type; };
template <typename T> struct ref_arr_traits<T *> { using value_type = T; };

template <typename A, typename T, typename Less>
ECAL_HD void ref_push_heap(A first, int64_t hole, int64_t top, T value, Less less) {   // std::__push_heap
    int64_t parent = (hole - 1) / 2;
    while (hole > top && less(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

template <typename A, typename T, typename Less>
ECAL_HD void ref_adjust_heap(A first, int64_t hole, int64_t len, T value, Less less) {   // std::__adjust_heap
    const int64_t top = hole;
    int64_t second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (less(first[second], first[second - 1])) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    ref_push_heap(first, hole, top, value, less);
}

template <typename A, typename Less>
ECAL_HD void ref_heap_select(A first, int64_t middle, int64_t last, Less less) {   // std::__heap_select(first, first + middle, first + last)
    using T = typename ref_arr_traits<A>::value_type;
    if (middle >= 2) {   // std::__make_heap(first, middle)
        for (int64_t parent = (middle - 2) / 2;; parent--) {
            const T value = first[parent];
            ref_adjust_heap(first, parent, middle, value, less);
            if (parent == 0) break;
        }
    }
    for (int64_t i = middle; i < last; i++)
        if (less(first[i], first[0])) {   // std::__pop_heap(first, middle, i)
            const T value = first[i];
            first[i] = first[0];
            ref_adjust_heap(first, (int64_t) 0, middle, value, less);
        }
}

// std::nth_element(a, a + nth, a + m, less); nth < m
template <typename A, typename Less>
ECAL_HD void ref_nth_element(A a, uint32_t m, uint32_t nth, Less less) {
    using T = typename ref_arr_traits<A>::value_type;
    if (m == 0 || nth >= m) return;
    auto swp = [&](uint32_t i, uint32_t j) {
        const T t = a[i];
        a[i] = a[j];
        a[j] = t;
    };
    uint32_t first = 0, last = m;
    uint32_t lg = 0;   // std::__lg(m)
    while ((m >> (lg + 1u)) != 0u) lg++;
    uint32_t depth = 2u * lg;
    while (last - first > 3u) {
        if (depth == 0u) {
            ref_heap_select(a + first, (int64_t) (nth + 1u - first), (int64_t) (last - first), less);
            swp(first, nth);
            return;
        }
        depth--;
        // __unguarded_partition_pivot
        const uint32_t mid = first + (last - first) / 2u;
        {   // __move_median_to_first(first, first + 1, mid, last - 1)
            const uint32_t pa = first + 1u, pb = mid, pc = last - 1u;
            if (less(a[pa], a[pb])) {
                if (less(a[pb], a[pc])) swp(first, pb);
                else if (less(a[pa], a[pc])) swp(first, pc);
                else swp(first, pa);
            } else if (less(a[pa], a[pc])) {
                swp(first, pa);
            } else if (less(a[pb], a[pc])) {
                swp(first, pc);
            } else {
                swp(first, pb);
            }
        }
        uint32_t lo = first + 1u, hi = last;   // __unguarded_partition(first + 1, last, pivot = first)
        for (;;) {
            while (less(a[lo], a[first])) lo++;
            hi--;
            while (less(a[first], a[hi])) hi--;
            if (!(lo < hi)) break;
            swp(lo, hi);
            lo++;
        }
        if (lo <= nth) first = lo;
        else last = lo;
    }
    // __insertion_sort(first, last)
    for (uint32_t i = first + 1u; i < last; i++) {
        const T val = a[i];
        if (less(val, a[first])) {
            for (uint32_t j = i; j > first; j--) a[j] = a[j - 1u];   // move_backward(first, i, i + 1)
            a[first] = val;
        } else {   // __unguarded_linear_insert
            uint32_t j = i;
            while (less(val, a[j - 1u])) {
                a[j] = a[j - 1u];
                j--;
            }
            a[j] = val;
        }
    }
}

}  // namespace ecal
