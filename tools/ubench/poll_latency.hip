// Microbenchmark (round 6): what a look at a counter another CU increments costs, by the path it takes -- for the cluster hand-offs of the
// 65536-point kernel.  Work-group 0 adds 1 to a counter every `gap` ticks (agent-scope atomic); one wave in each of the other work-groups
// polls it and records, per value, the ticks between the increment (the producer's clock reading, published beside the counter) and its own
// first sight of it, plus the round trip of a single poll.  Paths: 0 s_load_dword glc; 1 s_dcache_inv + s_load_dword (no glc);
// 2 s_load_dword without glc (expected: stale); 3 vector global_load_dword sc1 (what __hip_atomic_load relaxed / agent compiles to);
// 4 vector global_load_dword sc0 sc1.
// hipcc --offload-arch=gfx950 -O3 poll_latency.hip -o poll_latency && ./poll_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <int PATH>
static __device__ __forceinline__ uint32_t look(const uint32_t *p)
{
	uint32_t v;
	if (PATH == 0)      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
	else if (PATH == 1) asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
	else if (PATH == 2) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
	else if (PATH == 3) { uint32_t t; asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(0), "s"(p) : "memory"); v = __builtin_amdgcn_readfirstlane(t); }
	else                { uint32_t t; asm volatile("global_load_dword %0, %1, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(0), "s"(p) : "memory"); v = __builtin_amdgcn_readfirstlane(t); }
	return v;
}

template <int PATH>
__global__ __launch_bounds__(64) void k(uint32_t *cnt, long long *stamp, int n, int gap, long long *seen, long long *rtt)
{
	if (blockIdx.x == 0) {
		for (int i = 1; i <= n; i++) {
			const long long t0 = __builtin_readcyclecounter();
			while (__builtin_readcyclecounter() - t0 < gap) __builtin_amdgcn_s_sleep(1);
			if (threadIdx.x == 0) {
				__hip_atomic_store(stamp + i, (long long)__builtin_readcyclecounter(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		return;
	}
	uint32_t last = 0;
	long long rt_sum = 0; int rt_n = 0;
	for (long long spins = 0; last < (uint32_t)n && spins < (1ll << 24); spins++) {
		const long long a = __builtin_readcyclecounter();
		const uint32_t v = look<PATH>(cnt);
		const long long b = __builtin_readcyclecounter();
		rt_sum += b - a; rt_n++;
		if (v != last) {
			if (threadIdx.x == 0)
				for (uint32_t i = last + 1; i <= v && i <= (uint32_t)n; i++)
					seen[(size_t)blockIdx.x * (n + 1) + i] = b;
			last = v;
		}
	}
	if (threadIdx.x == 0) { rtt[blockIdx.x * 2] = rt_sum; rtt[blockIdx.x * 2 + 1] = rt_n; }
}

int main()
{
	const int n = 400, gap = 4000, wgs = 257;
	uint32_t *cnt; long long *stamp, *seen, *rtt;
	(void)hipMalloc(&cnt, 256); (void)hipMalloc(&stamp, sizeof(long long) * (n + 1));
	(void)hipMalloc(&seen, sizeof(long long) * wgs * (n + 1)); (void)hipMalloc(&rtt, sizeof(long long) * wgs * 2);
	const char *names[] = { "s_load_dword glc", "s_dcache_inv + s_load_dword", "s_load_dword (no glc)", "global_load_dword sc1", "global_load_dword sc0 sc1" };
	for (int path = 0; path < 5; path++) {
		(void)hipMemset(cnt, 0, 256); (void)hipMemset(seen, 0, sizeof(long long) * wgs * (n + 1)); (void)hipMemset(rtt, 0, sizeof(long long) * wgs * 2);
		switch (path) {
#define C(P) case P: hipLaunchKernelGGL(k<P>, dim3(wgs), dim3(64), 0, 0, cnt, stamp, n, gap, seen, rtt); break;
		C(0) C(1) C(2) C(3) C(4)
		}
		if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", names[path]); return 1; }
		std::vector<long long> hs(n + 1), hseen((size_t)wgs * (n + 1)), hr(wgs * 2);
		(void)hipMemcpy(hs.data(), stamp, sizeof(long long) * (n + 1), hipMemcpyDeviceToHost);
		(void)hipMemcpy(hseen.data(), seen, sizeof(long long) * wgs * (n + 1), hipMemcpyDeviceToHost);
		(void)hipMemcpy(hr.data(), rtt, sizeof(long long) * wgs * 2, hipMemcpyDeviceToHost);
		std::vector<double> lat; double rt = 0; long long rn = 0; long long missed = 0;
		for (int w = 1; w < wgs; w++) {
			rt += (double)hr[w * 2]; rn += hr[w * 2 + 1];
			for (int i = 10; i <= n; i++) {
				const long long s = hseen[(size_t)w * (n + 1) + i];
				if (!s) { missed++; continue; }
				lat.push_back((double)(s - hs[i]));
			}
		}
		std::sort(lat.begin(), lat.end());
		if (lat.empty()) { printf("%-30s nothing seen (%lld missed)\n", names[path], missed); continue; }
		printf("%-30s round trip %.0f ticks; increment -> first sight: median %.0f, p90 %.0f, max %.0f ticks; %lld values never seen (256 pollers, s_memtime ticks)\n",
		       names[path], rt / (double)rn, lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat.back(), missed);
	}
	return 0;
}
