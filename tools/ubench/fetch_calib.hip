// Calibration of rocprofv3's FETCH_SIZE on gfx950 per access width: each kernel streams the same 1 GiB buffer once
// (more than the 256 MiB Infinity Cache) with 4-, 8- or 16-byte loads per lane, plain or non-temporal.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d out -o p -- ./fetch_calib
// and compare FETCH_SIZE (reported in KiB-like units of the tool; x 1024 = bytes per its own summary) with 2^30 bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename T, bool NT>
__global__ __launch_bounds__(256) void stream(const T *__restrict__ src, float *out, size_t n)
{
	float acc = 0.0f;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		T v = NT ? __builtin_nontemporal_load(src + i) : src[i];
		acc += reinterpret_cast<const float *>(&v)[0];
	}
	if (acc == 123.456f) out[0] = acc;
}

#define RUN(T, NT, name) do { \
	hipLaunchKernelGGL((stream<T, NT>), dim3(4096), dim3(256), 0, 0, (const T *)d, out, bytes / sizeof(T)); \
	hipDeviceSynchronize(); printf("%s done\n", name); } while (0)

int main()
{
	const size_t bytes = (size_t)1 << 30;
	void *d; float *out;
	hipMalloc(&d, bytes); hipMalloc(&out, 4);
	hipMemset(d, 0, bytes);
	hipDeviceSynchronize();
	for (int rep = 0; rep < 2; rep++) {
		RUN(float, false, "b4"); RUN(float, true, "b4_nt");
		RUN(v2f, false, "b8");   RUN(v2f, true, "b8_nt");
		RUN(v4f, false, "b16");  RUN(v4f, true, "b16_nt");
	}
	return 0;
}
