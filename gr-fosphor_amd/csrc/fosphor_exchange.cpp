/*
 * fosphor_exchange.cpp -- the per-display-frame exchange of the multi-GPU split, native (RCCL over xGMI)
 *
 * after the time-sharded FFT + count every rank holds partial arrays whose combination is three
 * commutative reductions -- hit counts (uint32, sum: exact and order-independent, so bit-identical on every
 * rank and to the single-GPU result), live sum (float, sum), max (float, max).  They go out as ONE group
 * (ncclGroupStart / ncclGroupEnd) on the instance's count/merge stream, between K2 and K3, so that the
 * exchange of frame k overlaps K1 of frame k + 1 without any host synchronisation.
 *
 * RCCL is bound at run time (dlopen "librccl.so.1"): a process that already carries an RCCL -- PyTorch ships
 * its own copy -- keeps exactly one, and a single-GPU user of libfosphor_amd.so needs none.
 */
#include <dlfcn.h>
#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "fosphor_internal.h"

/* The slice of the NCCL / RCCL C ABI this file binds with dlsym, declared here so that the library builds where no
 * RCCL headers are installed (a single-GPU user needs neither headers nor library).  Values are the public ABI of
 * nccl.h / rccl.h (stable since NCCL 2.x; checked against /opt/rocm/include/rccl/rccl.h: ncclSuccess 0, ncclSum 0,
 * ncclMax 2, ncclUint32 3, ncclFloat32 7, 128-byte unique id). */
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
typedef enum { ncclUint32 = 3, ncclFloat32 = 7 } ncclDataType_t;
}

namespace fosphor_amd {

struct Rccl {
	void *lib;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *);
	ncclResult_t (*CommInitRank)(ncclComm_t *, int nranks, ncclUniqueId id, int rank);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*CommCount)(const ncclComm_t, int *);
	ncclResult_t (*GroupStart)(void);
	ncclResult_t (*GroupEnd)(void);
	ncclResult_t (*AllReduce)(const void *send, void *recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
	ncclResult_t (*ReduceScatter)(const void *send, void *recv, size_t recvcount, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
	ncclResult_t (*AllGather)(const void *send, void *recv, size_t sendcount, ncclDataType_t, ncclComm_t, hipStream_t);
	const char  *(*GetErrorString)(ncclResult_t);
};

static Rccl *rccl(void)
{
	static Rccl r;
	static int state;			/* 0 untried, 1 ok, -1 unavailable */
	static std::mutex mu;
	std::lock_guard<std::mutex> g(mu);
	if (state == 0) {
		const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
		for (const char *n : names) {
			r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
			if (r.lib)
				break;
		}
		state = -1;
		if (!r.lib) {
			fprintf(stderr, "[!] fosphor_amd: RCCL not found (%s): the multi-GPU exchange is unavailable\n", dlerror());
		} else {
#define BIND(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym))
			BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank");
			BIND(CommDestroy, "ncclCommDestroy"); BIND(GroupStart, "ncclGroupStart"); BIND(GroupEnd, "ncclGroupEnd");
			BIND(AllReduce, "ncclAllReduce"); BIND(ReduceScatter, "ncclReduceScatter");
			BIND(AllGather, "ncclAllGather"); BIND(GetErrorString, "ncclGetErrorString"); BIND(CommCount, "ncclCommCount");
#undef BIND
			if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.AllReduce &&
			    r.ReduceScatter && r.AllGather)
				state = 1;
			else
				fprintf(stderr, "[!] fosphor_amd: the RCCL library lacks an entry point\n");
		}
	}
	return state == 1 ? &r : nullptr;
}

static int check(Rccl *r, ncclResult_t e, const char *what)
{
	if (e == ncclSuccess)
		return 0;
	fprintf(stderr, "[!] fosphor_amd: %s: %s\n", what, r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
	return -EIO;
}

int xchg_available(void)
{
	return rccl() ? 1 : 0;
}

/* ranks of the communicator as RCCL itself counts them */
int xchg_comm_count(void *comm)
{
	Rccl *r = rccl();
	int n = 0;
	if (!r || !r->CommCount)
		return -ENOSYS;
	if (check(r, r->CommCount((ncclComm_t)comm, &n), "ncclCommCount"))
		return -EIO;
	return n;
}

int xchg_unique_id(void *id128)
{
	Rccl *r = rccl();
	ncclUniqueId id;
	if (!r)
		return -ENOSYS;
	if (check(r, r->GetUniqueId(&id), "ncclGetUniqueId"))
		return -EIO;
	memcpy(id128, &id, sizeof(id));
	return 0;
}

int xchg_comm_init(void **comm, int world, int rank, const void *id128)
{
	Rccl *r = rccl();
	ncclUniqueId id;
	ncclComm_t c = nullptr;
	if (!r)
		return -ENOSYS;
	memcpy(&id, id128, sizeof(id));
	if (check(r, r->CommInitRank(&c, world, id, rank), "ncclCommInitRank"))
		return -EIO;
	*comm = c;
	return 0;
}

int xchg_comm_destroy(void *comm)
{
	Rccl *r = rccl();
	if (!r)
		return -ENOSYS;
	return check(r, r->CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}

/* one group: AllReduce(hc, u32, sum); AllReduce(S, f32, sum); AllReduce(max, f32, max) -- in place */
int xchg_allreduce3(void *comm, hipStream_t st, uint32_t *hc, size_t n_hc, float *sum, float *mx, size_t n_cols)
{
	Rccl *r = rccl();
	ncclComm_t c = (ncclComm_t)comm;
	int rv = 0;
	if (!r)
		return -ENOSYS;
	if (check(r, r->GroupStart(), "ncclGroupStart"))
		return -EIO;
	rv |= check(r, r->AllReduce(hc, hc, n_hc, ncclUint32, ncclSum, c, st), "ncclAllReduce(hit counts)");
	rv |= check(r, r->AllReduce(sum, sum, n_cols, ncclFloat32, ncclSum, c, st), "ncclAllReduce(live sum)");
	rv |= check(r, r->AllReduce(mx, mx, n_cols, ncclFloat32, ncclMax, c, st), "ncclAllReduce(max)");
	rv |= check(r, r->GroupEnd(), "ncclGroupEnd");
	return rv ? -EIO : 0;
}

/* frequency-sliced form for large states: the hit counts are reduce-SCATTERED (rank r receives the sums of
 * cells [r n_hc / world, (r + 1) n_hc / world), in place), the two column arrays all-reduced */
int xchg_reduce_scatter(void *comm, hipStream_t st, uint32_t *hc, size_t n_hc, int world, int rank,
                        float *sum, float *mx, size_t n_cols)
{
	Rccl *r = rccl();
	ncclComm_t c = (ncclComm_t)comm;
	const size_t per = n_hc / (size_t)world;
	int rv = 0;
	if (!r)
		return -ENOSYS;
	if (check(r, r->GroupStart(), "ncclGroupStart"))
		return -EIO;
	rv |= check(r, r->ReduceScatter(hc, hc + per * rank, per, ncclUint32, ncclSum, c, st), "ncclReduceScatter(hit counts)");
	rv |= check(r, r->AllReduce(sum, sum, n_cols, ncclFloat32, ncclSum, c, st), "ncclAllReduce(live sum)");
	rv |= check(r, r->AllReduce(mx, mx, n_cols, ncclFloat32, ncclMax, c, st), "ncclAllReduce(max)");
	rv |= check(r, r->GroupEnd(), "ncclGroupEnd");
	return rv ? -EIO : 0;
}

/* every rank's slice of a float array to every rank, in place */
int xchg_allgather_f32(void *comm, hipStream_t st, float *a, size_t n, int world, int rank)
{
	Rccl *r = rccl();
	const size_t per = n / (size_t)world;
	if (!r)
		return -ENOSYS;
	return check(r, r->AllGather(a + per * rank, a, per, ncclFloat32, (ncclComm_t)comm, st), "ncclAllGather(histogram)");
}

} // namespace fosphor_amd
