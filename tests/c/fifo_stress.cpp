// CPU-only stress of fosphor_amd::fifo (include/fosphor_amd_sink.h): one producer, one consumer, random region sizes, the
// consumer checks that every sample arrives once and in order.  Built with -fsanitize=thread by tests/test_boundary_cpu.py
// (the ring's two sides share only two atomic counters and a sleep mutex; this is what checks that).
//   g++ -std=c++17 -O1 -g -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fifo_stress.cpp ../../gr-fosphor_amd/csrc/fosphor_sink.cpp \
//       -L../../gr-fosphor_amd -lfosphor_amd -L/opt/rocm/lib -lamdhip64 -pthread
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include "../../include/fosphor_amd_sink.h"

using fosphor_amd::fifo;

int main(int argc, char **argv)
{
	const long total = argc > 1 ? atol(argv[1]) : 20000000L;
	fifo f(1 << 12, false);
	long bad = 0;
	std::thread prod([&] {
		unsigned s = 12345;
		long n = 0;
		while (n < total) {
			s = s * 1664525u + 1013904223u;
			int want = 1 + (int)((s >> 8) % 700);
			int mw = f.write_max_size();
			if (want > mw) want = mw;
			if (want > total - n) want = (int)(total - n);
			std::complex<float> *p = (s & 1) ? f.write_prepare(want, true) : f.write_prepare_for(want, 50);
			if (!p) continue;
			for (int i = 0; i < want; i++)
				p[i] = std::complex<float>((float)((n + i) & 0xffffff), (float)((n + i) >> 24));
			f.write_commit(want);
			n += want;
		}
	});
	std::thread cons([&] {
		unsigned s = 777;
		long n = 0;
		while (n < total) {
			s = s * 1664525u + 1013904223u;
			int want = 1 + (int)((s >> 8) % 900);
			int mr = f.read_max_size();
			if (want > mr) want = mr;
			if (want > total - n) want = (int)(total - n);
			std::complex<float> *p = f.read_peek(want, (s & 2) != 0);
			if (!p) { std::this_thread::yield(); continue; }
			if (f.peek_max_size_at(0) < want) bad++;
			for (int i = 0; i < want; i++)
				if (p[i].real() != (float)((n + i) & 0xffffff) || p[i].imag() != (float)((n + i) >> 24))
					bad++;
			f.read_discard(want);
			n += want;
		}
	});
	prod.join();
	cons.join();
	printf("fifo stress: %ld samples, %ld bad, used %d free %d\n", total, bad, f.used(), f.free());
	return (bad || f.used() != 0) ? 1 : 0;
}
