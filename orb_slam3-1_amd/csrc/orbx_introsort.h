// orbx_introsort.h -- reproduces, element move for element move, what libstdc++'s std::sort does, so
// that the device-side octree expands nodes in exactly the order the reference does.
//
// Why: DistributeOctTree sorts (keyCount, node*) pairs with std::sort and a comparator that ties on
// (count, UL.x) (reference src/ORBextractor.cc:538-553,700).  std::sort is not stable, so the order of
// tied elements -- and therefore which nodes get split before the feature budget is hit and the final
// keypoint order -- is defined by the libstdc++ algorithm: introsort (median-of-3 quicksort with an
// unguarded Hoare partition, depth limit 2*floor(log2 n), heapsort fallback) followed by a final
// insertion sort with threshold 16.  This file restates that published algorithm (GCC bits/stl_algo.h,
// bits/stl_heap.h) iteratively (explicit stack instead of recursion) for host and device.
// tests/test_introsort.py checks it against std::sort itself on tie-heavy and adversarial inputs.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define ORBX_SORT_HD __host__ __device__
#else
#define ORBX_SORT_HD
#endif

namespace orbx {

struct SortNode {
    int32_t count;   // pair.first  (number of keys in the node)
    int32_t ulx;     // pair.second->UL.x
    int32_t node;    // pair.second (node slot)
};

// compareNodes (reference src/ORBextractor.cc:538-553)
ORBX_SORT_HD inline bool node_less(const SortNode& a, const SortNode& b)
{
    if (a.count < b.count) return true;
    if (a.count > b.count) return false;
    return a.ulx < b.ulx;
}

namespace detail {

ORBX_SORT_HD inline void swap_nodes(SortNode* v, int a, int b)
{
    SortNode t = v[a]; v[a] = v[b]; v[b] = t;
}

ORBX_SORT_HD inline int floor_log2(int n)
{
    int r = 0;
    while (n > 1) { n >>= 1; ++r; }
    return r;
}

// __push_heap on v[first..], hole/top indices relative to first
ORBX_SORT_HD inline void push_heap(SortNode* v, int first, int hole, int top, SortNode value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && node_less(v[first + parent], value)) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}

// __adjust_heap
ORBX_SORT_HD inline void adjust_heap(SortNode* v, int first, int hole, int len, SortNode value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (node_less(v[first + child], v[first + child - 1])) child--;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + child - 1];
        hole = child - 1;
    }
    push_heap(v, first, hole, top, value);
}

// __partial_sort(first, last, last) == make_heap + sort_heap on [first, last)
ORBX_SORT_HD inline void heap_sort(SortNode* v, int first, int last)
{
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            SortNode value = v[first + parent];
            adjust_heap(v, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        SortNode value = v[last];
        v[last] = v[first];
        adjust_heap(v, first, 0, last - first, value);
    }
}

// __unguarded_linear_insert
ORBX_SORT_HD inline void unguarded_linear_insert(SortNode* v, int last)
{
    SortNode val = v[last];
    int next = last - 1;
    while (node_less(val, v[next])) {
        v[last] = v[next];
        last = next;
        --next;
    }
    v[last] = val;
}

// __insertion_sort
ORBX_SORT_HD inline void insertion_sort(SortNode* v, int first, int last)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (node_less(v[i], v[first])) {
            SortNode val = v[i];
            for (int k = i; k > first; --k) v[k] = v[k - 1];
            v[first] = val;
        } else {
            unguarded_linear_insert(v, i);
        }
    }
}

}  // namespace detail

// std::sort(v, v+n, compareNodes)
// `stack` = 3 * kIntrosortStack ints of caller-provided storage (LDS on the device: a private array would live in scratch memory)
constexpr int kIntrosortStack = 64;
ORBX_SORT_HD inline void introsort_nodes(SortNode* v, int n, int* stack)
{
    using namespace detail;
    if (n <= 0) return;
    // __introsort_loop with an explicit stack of (first, last, depth) for the recursive right halves
    int* const stack_first = stack; int* const stack_last = stack + kIntrosortStack; int* const stack_depth = stack + 2 * kIntrosortStack;
    int sp = 0;
    stack_first[0] = 0; stack_last[0] = n; stack_depth[0] = floor_log2(n) * 2; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
        // the recursion visits the RIGHT part first (call), then loops on the left part; the two parts are
        // disjoint so the visiting order does not change the result -- process left now, push right.
        while (last - first > 16) {
            if (depth == 0) { heap_sort(v, first, last); break; }
            --depth;
            // __unguarded_partition_pivot
            const int mid = first + (last - first) / 2;
            const int a = first + 1, b = mid, c = last - 1;
            if (node_less(v[a], v[b])) {
                if (node_less(v[b], v[c])) swap_nodes(v, first, b);
                else if (node_less(v[a], v[c])) swap_nodes(v, first, c);
                else swap_nodes(v, first, a);
            } else if (node_less(v[a], v[c])) swap_nodes(v, first, a);
            else if (node_less(v[b], v[c])) swap_nodes(v, first, c);
            else swap_nodes(v, first, b);
            int lo = first + 1, hi = last;
            while (true) {
                while (node_less(v[lo], v[first])) ++lo;
                --hi;
                while (node_less(v[first], v[hi])) --hi;
                if (!(lo < hi)) break;
                swap_nodes(v, lo, hi);
                ++lo;
            }
            const int cut = lo;
            if (sp < kIntrosortStack) { stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth; ++sp; }
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        insertion_sort(v, 0, 16);
        for (int i = 16; i != n; ++i) unguarded_linear_insert(v, i);
    } else {
        insertion_sort(v, 0, n);
    }
}


#if defined(__HIPCC__)
// The same std::sort, run by one 64-lane wave (the only wave of its workgroup) instead of by one lane.
//
// introsort's result is fixed by two things: the sequence of Hoare partitions (which elements get swapped) and the final
// insertion sort.  Both have data-parallel restatements that move exactly the same elements:
//  * __unguarded_partition(first + 1, last, pivot = v[first]): the left cursor only ever stops on elements that are
//    not < pivot and the right cursor on elements the pivot is not <; between two swaps neither cursor crosses an element
//    the other has touched, so the k-th stop of the left cursor is the k-th such position in ascending order of the
//    UNSWAPPED array (L_k) and the k-th stop of the right cursor the k-th such position in descending order (R_k).  The loop
//    swaps (L_k, R_k) while L_k < R_k -- K swaps -- and returns min(L_{K+1}, R_K).  One ballot pass ranks the stops, one pass
//    swaps the K pairs.
//  * __final_insertion_sort is a stable sort, and after the partition phase every leaf segment (<= 16 elements, or a
//    heap-sorted one) is already in its final range: each element's final place is its segment's start plus its stable
//    rank inside the segment (<= 16 comparisons).
// `out` receives the sorted array; it doubles as scratch while partitioning (3 ints per element: left stops, right stops,
// segment bounds).  `rank_tmp`: n shorts.  `stack`: 3 * kIntrosortStack ints.  All four live in the caller's LDS (or an HBM
// scratch); every lane of the wave must call this with the same arguments.
__device__ inline void wave_sort_nodes(SortNode* v, SortNode* out, int n, int* stack, short* rank_tmp)
{
    using namespace detail;
    if (n <= 0) return;
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int* const posL = (int*)out; int* const posR = posL + n; int* const seg = posR + n;
    int* const stack_first = stack; int* const stack_last = stack + kIntrosortStack; int* const stack_depth = stack + 2 * kIntrosortStack;
    int sp = 0;
    int first = 0, last = n, depth = floor_log2(n) * 2;
    while (true) {
        while (last - first > 16) {
            if (depth == 0) {           // __partial_sort(first, last, last): heapsort, one lane (adversarial inputs only)
                if (lane == 0) heap_sort(v, first, last);
                __syncthreads();
                break;
            }
            --depth;
            // __move_median_to_first(first, first + 1, mid, last - 1)
            const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
            const SortNode A = v[ia], Bm = v[ib], Cc = v[ic], F = v[first];
            int ch;
            if (node_less(A, Bm)) ch = node_less(Bm, Cc) ? ib : (node_less(A, Cc) ? ic : ia);
            else ch = node_less(A, Cc) ? ia : (node_less(Bm, Cc) ? ic : ib);
            const SortNode piv = (ch == ia) ? A : ((ch == ib) ? Bm : Cc);
            // rank the cursor stops on the array as it is after that swap (position ch now holds the old v[first])
            int nL = 0, nR = 0;
            for (int p0 = first + 1; p0 < last; p0 += 64) {
                const int p = p0 + lane;
                bool ge = false, le = false;
                if (p < last) {
                    SortNode e = v[p];
                    if (p == ch) e = F;
                    ge = !node_less(e, piv); le = !node_less(piv, e);
                }
                const unsigned long long mg = __ballot(ge), ml = __ballot(le);
                if (ge) posL[nL + __popcll(mg & lt)] = p;
                if (le) posR[nR + __popcll(ml & lt)] = p;           // ascending; the right cursor's k-th stop is posR[nR - 1 - k]
                nL += __popcll(mg); nR += __popcll(ml);
            }
            __syncthreads();
            if (lane == 0) { v[first] = piv; v[ch] = F; }
            __syncthreads();
            // K = number of swaps; the pairs are monotone (L ascending, R descending), so the test holds on a prefix
            int K = 0, cutL = 0x7fffffff, cutR = 0x7fffffff;
            const int kmax = nL < nR ? nL : nR;
            for (int k0 = 0; k0 < kmax; k0 += 64) {
                const int k = k0 + lane;
                int a = 0, b = 0;
                if (k < kmax) { a = posL[k]; b = posR[nR - 1 - k]; }
                const unsigned long long m = __ballot(k < kmax && a < b);
                K += __popcll(m);
                if (m != ~0ull) break;
            }
            if (K < nL) cutL = posL[K];
            if (K >= 1) cutR = posR[nR - K];
            const int cut = cutL < cutR ? cutL : cutR;
            for (int k0 = 0; k0 < K; k0 += 64) {
                const int k = k0 + lane;
                if (k < K) {
                    const int a = posL[k], b = posR[nR - 1 - k];
                    const SortNode x = v[a], y = v[b];
                    v[a] = y; v[b] = x;
                }
            }
            __syncthreads();
            // __introsort_loop(cut, last, depth) -- later; continue with [first, cut)
            if (sp < kIntrosortStack) { stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth; ++sp; }
            last = cut;
        }
        // [first, last) is a leaf: every element's final place lies inside it
        for (int p = first + lane; p < last; p += 64) seg[p] = first | (last << 16);
        if (sp == 0) break;
        __syncthreads();
        --sp;
        first = stack_first[sp]; last = stack_last[sp]; depth = stack_depth[sp];
    }
    __syncthreads();
    // __final_insertion_sort: stable rank inside the leaf
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        int lo = 0, len = 0, rank = 0;
        SortNode me; me.count = 0; me.ulx = 0; me.node = 0;
        if (i < n) { const int sg = seg[i]; lo = sg & 0xFFFF; len = (sg >> 16) - lo; me = v[i]; }
        for (int t = 0; __ballot(t < len) != 0ull; t++) {
            if (t < len) {
                const int j = lo + t;
                const SortNode o = v[j];
                const bool before = node_less(o, me) || (!node_less(me, o) && j < i);
                rank += before ? 1 : 0;
            }
        }
        if (i < n) rank_tmp[i] = (short)(lo + rank);
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) out[rank_tmp[i]] = v[i];
    __syncthreads();
}
#endif

}  // namespace orbx
