// check rocPRIM scans / radix sort beyond 2^32 items (diagnostic, not product code)
#include <cstdio>
#include <cstdint>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
struct U8 { __device__ uint64_t operator()(uint8_t v) const { return v; } };
struct MaxOp { __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; } };
__global__ void fill(uint8_t *f, uint64_t *h, uint64_t n) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { f[i] = (i % 3 == 0); h[i] = (i % 1000 == 0) ? i : 0; } }
__global__ void keys(uint64_t *k, uint64_t *v, uint64_t n) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { k[i] = i * 0x9E3779B97F4A7C15ull; v[i] = i; } }
__global__ void check_sorted(const uint64_t *k, const uint64_t *v, uint64_t n, unsigned long long *bad) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i + 1 < n && k[i] > k[i + 1]) atomicAdd(bad, 1ull); if (i < n && k[i] != v[i] * 0x9E3779B97F4A7C15ull) atomicAdd(bad + 1, 1ull); }
int main() {
	uint64_t n = 5000000001ull;
	uint8_t *f; uint64_t *h, *pos; void *tmp = nullptr; size_t tb = 0;
	hipMalloc(&f, n); hipMalloc(&h, n * 8); hipMalloc(&pos, n * 8);
	fill<<<(unsigned)((n + 255) / 256), 256>>>(f, h, n);
	auto in = rocprim::make_transform_iterator(f, U8());
	rocprim::exclusive_scan(nullptr, tb, in, pos, (uint64_t)0, n, rocprim::plus<uint64_t>());
	hipMalloc(&tmp, tb);
	rocprim::exclusive_scan(tmp, tb, in, pos, (uint64_t)0, n, rocprim::plus<uint64_t>());
	uint64_t last = 0; hipMemcpy(&last, pos + n - 1, 8, hipMemcpyDeviceToHost);
	printf("exclusive_scan: last = %llu expect %llu\n", (unsigned long long)last, (unsigned long long)((n - 1 + 2) / 3));
	hipFree(tmp); tb = 0;
	rocprim::inclusive_scan(nullptr, tb, h, h, n, MaxOp());
	hipMalloc(&tmp, tb);
	rocprim::inclusive_scan(tmp, tb, h, h, n, MaxOp());
	hipMemcpy(&last, h + n - 1, 8, hipMemcpyDeviceToHost);
	printf("inclusive max scan: last = %llu expect %llu\n", (unsigned long long)last, (unsigned long long)((n - 1) / 1000 * 1000));
	uint64_t mid = 0; hipMemcpy(&mid, h + 4400000123ull, 8, hipMemcpyDeviceToHost);
	printf("inclusive max scan: [4400000123] = %llu expect 4400000000\n", (unsigned long long)mid);
	hipFree(tmp); hipFree(f); hipFree(pos); hipFree(h);
	uint64_t *k0, *k1, *v0, *v1; unsigned long long *bad;
	hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 8); hipMalloc(&v1, n * 8); hipMalloc(&bad, 16); hipMemset(bad, 0, 16);
	keys<<<(unsigned)((n + 255) / 256), 256>>>(k0, v0, n);
	rocprim::double_buffer<uint64_t> kb(k0, k1), vb(v0, v1);
	tb = 0; rocprim::radix_sort_pairs(nullptr, tb, kb, vb, (size_t)n, 0u, 64u);
	hipMalloc(&tmp, tb);
	hipError_t e = rocprim::radix_sort_pairs(tmp, tb, kb, vb, (size_t)n, 0u, 64u);
	hipDeviceSynchronize();
	check_sorted<<<(unsigned)((n + 255) / 256), 256>>>(kb.current(), vb.current(), n, bad);
	unsigned long long hb[2]; hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
	printf("radix_sort_pairs: err %d temp %zu bytes, unsorted pairs %llu, wrong values %llu, last error %s\n", (int)e, tb, hb[0], hb[1], hipGetErrorString(hipGetLastError()));
	return 0;
}
