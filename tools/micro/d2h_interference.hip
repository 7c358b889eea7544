// tools/micro/d2h_interference.hip -- what does a device-to-host copy at link speed do to (a) the latency of a host round trip
// (tiny kernel + hipStreamSynchronize) and (b) the speed of a latency-bound gather kernel?  (host pipeline)
//   hipcc --offload-arch=gfx950 -O3 -o d2h_interference d2h_interference.hip && ./d2h_interference
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void tiny(unsigned long long *dst, const unsigned long long *src) { if (threadIdx.x < 32) dst[threadIdx.x] = src[threadIdx.x]; __threadfence_system(); }
__global__ void gather(const uint4 *tab, size_t n, int steps, unsigned long long *sink)
{
	unsigned long long k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
	for (int i = 0; i < steps; ++i) { uint4 v = tab[(k >> 11) % n]; k = k * 6364136223846793005ull + v.x + v.w + 1442695040888963407ull; }
	if (k == 42) *sink = k;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t COPY = (size_t)3 << 30, TAB = (size_t)8 << 30;
	hipStream_t sc, sd; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
	void *d_src, *h_dst; uint4 *tab; unsigned long long *d_ctr, *h_ctr, *h_ctr_dev, *sink;
	CK(hipMalloc(&d_src, COPY)); CK(hipHostMalloc(&h_dst, COPY, hipHostMallocDefault)); CK(hipMalloc((void **)&tab, TAB)); CK(hipMemset(tab, 1, TAB));
	CK(hipMalloc((void **)&d_ctr, 256)); CK(hipMemset(d_ctr, 0, 256)); CK(hipHostMalloc((void **)&h_ctr, 256, hipHostMallocDefault)); CK(hipMalloc((void **)&sink, 8));
	CK(hipHostGetDevicePointer((void **)&h_ctr_dev, h_ctr, 0));
	CK(hipDeviceSynchronize());
	auto round_trips = [&](int n, bool by_kernel) { // mean latency of n host round trips, microseconds
		double t0 = now_ms();
		for (int i = 0; i < n; ++i) {
			if (by_kernel) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sc, h_ctr_dev, d_ctr);
			else CK(hipMemcpyAsync(h_ctr, d_ctr, 256, hipMemcpyDeviceToHost, sc));
			CK(hipStreamSynchronize(sc));
		}
		return (now_ms() - t0) * 1e3 / n;
	};
	auto gather_ms = [&]() {
		double t0 = now_ms();
		hipLaunchKernelGGL(gather, dim3(256 * 8), dim3(256), 0, sc, tab, TAB / 16, 400, sink);
		CK(hipStreamSynchronize(sc));
		return now_ms() - t0;
	};
	round_trips(20, true); gather_ms();
	printf("idle:        round trip by kernel store %.1f us, by memcpy %.1f us; gather kernel %.2f ms\n", round_trips(200, true), round_trips(200, false), gather_ms());
	for (size_t piece : {COPY, (size_t)64 << 20, (size_t)8 << 20, (size_t)1 << 20}) {
		for (int what = 0; what < 3; ++what) {
			double t0 = now_ms();
			for (size_t o = 0; o < COPY; o += piece) CK(hipMemcpyAsync((char *)h_dst + o, (char *)d_src + o, piece, hipMemcpyDeviceToHost, sd));
			double issued = now_ms() - t0, v = 0;
			if (what == 0) v = round_trips(40, true); else if (what == 1) v = round_trips(40, false); else v = gather_ms();
			double mid = now_ms() - t0;
			CK(hipStreamSynchronize(sd));
			double all = now_ms() - t0;
			printf("D2H 3 GiB in %4zu MiB pieces (issue %.1f ms, done %.1f ms = %.1f GB/s): %s %.1f %s (measured over the first %.1f ms)\n", piece >> 20, issued, all, COPY / all / 1e6,
			       what == 0 ? "round trip by kernel store" : what == 1 ? "round trip by memcpy" : "gather kernel", v, what == 2 ? "ms" : "us", mid);
		}
	}
	return 0;
}
