// rccl_standin.cpp -- TEST INFRASTRUCTURE, never shipped in substrata_amd/: a stand-in for the handful of RCCL entry points libsgp.so binds
// (sgp_world_tiles.hip: ncclGetUniqueId / CommInitRank / CommDestroy / CommAbort / AllGather / Send / Recv / GroupStart / GroupEnd / GetErrorString /
// CommCount), so that SEVERAL OS PROCESSES ON ONE GPU can run sgp_tiles_exchange itself -- the product's own routing, gather, send / receive and import
// code, not a restatement of it -- on the one-GPU boxes the tests get (real RCCL wants one device per rank).  Injected with SGP_RCCL_LIBRARY.
//
// Transport: files under /dev/shm/sgp_rccl_standin_<unique id>/ (host staging: device -> host copy -> file -> host -> device copy).  Every operation
// completes before its call returns (legal for a stream-ordered API: the stream is drained first), grouped sends are all written before any
// grouped receive is awaited (so no cycle of waits), a receive that finds nothing within SGP_RCCL_STANDIN_TIMEOUT_S (default 30 s) returns
// ncclSystemError: a protocol bug in the exchange shows up as a failed test, not as a hang.  Nothing measured through this library is a scaling
// number: tests label it "transport": "test stand-in".
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <fcntl.h>
#include <unistd.h>
#include <dirent.h>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
struct ncclComm {
	std::string dir;
	int nranks = 0, rank = 0;
	uint64_t ag_seq = 0;
	std::vector<uint64_t> send_seq, recv_seq;      // per peer
	bool aborted = false;
};
typedef struct ncclComm* ncclComm_t;
}

namespace {
struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_group_ops;

size_t type_size(ncclDataType_t t) { switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; } }
double timeout_s() { const char* e = getenv("SGP_RCCL_STANDIN_TIMEOUT_S"); return e && *e ? atof(e) : 30.0; }

bool write_file(const std::string& path, const void* data, size_t bytes)
{
	const std::string tmp = path + ".tmp";
	FILE* f = fopen(tmp.c_str(), "wb");
	if (!f) return false;
	const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
	fclose(f);
	return ok && rename(tmp.c_str(), path.c_str()) == 0;      // (atomic: a reader sees the whole message or none)
}
// waits for the file, reads exactly `bytes`; false on timeout, abort or a size mismatch
bool read_file(ncclComm_t c, const std::string& path, void* data, size_t bytes, bool unlink_after)
{
	const auto t0 = std::chrono::steady_clock::now();
	const std::string abort_flag = c->dir + "/abort";
	for (int spin = 0;; ++spin) {
		struct stat st;
		if (stat(path.c_str(), &st) == 0) {
			if ((size_t)st.st_size != bytes) { fprintf(stderr, "[rccl stand-in] %s: %zu bytes where %zu were expected (send / recv counts disagree)\n", path.c_str(), (size_t)st.st_size, bytes); return false; }
			FILE* f = fopen(path.c_str(), "rb");
			if (!f) return false;
			const bool ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
			fclose(f);
			if (unlink_after) unlink(path.c_str());
			return ok;
		}
		if (c->aborted || access(abort_flag.c_str(), F_OK) == 0) return false;
		if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
			fprintf(stderr, "[rccl stand-in] rank %d: nothing at %s after %.0f s (a peer never sent it)\n", c->rank, path.c_str(), timeout_s());
			return false;
		}
		if (spin < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
	}
}

ncclResult_t do_send(const Op& op)
{
	std::vector<char> host(op.bytes);
	if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
	ncclComm_t c = op.comm;
	char name[96]; snprintf(name, sizeof(name), "/p2p_%d_%d_%llu", c->rank, op.peer, (unsigned long long)c->send_seq[op.peer]++);
	return write_file(c->dir + name, host.data(), op.bytes) ? ncclSuccess : ncclSystemError;
}
ncclResult_t do_recv(const Op& op)
{
	ncclComm_t c = op.comm;
	std::vector<char> host(op.bytes);
	char name[96]; snprintf(name, sizeof(name), "/p2p_%d_%d_%llu", op.peer, c->rank, (unsigned long long)c->recv_seq[op.peer]++);
	if (!read_file(c, c->dir + name, host.data(), op.bytes, true)) return ncclSystemError;
	if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
	return ncclSuccess;
}
ncclResult_t run_ops(std::vector<Op>& ops)
{
	// what the stream holds so far produced the send buffers: drain it first
	for (const Op& op : ops) if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
	for (const Op& op : ops) if (op.send) { const ncclResult_t r = do_send(op); if (r != ncclSuccess) return r; }
	for (const Op& op : ops) if (!op.send) { const ncclResult_t r = do_recv(op); if (r != ncclSuccess) return r; }
	return ncclSuccess;
}
}

extern "C" {
#define STANDIN_API __attribute__((visibility("default")))

STANDIN_API ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
	if (!id) return ncclInvalidArgument;
	memset(id, 0, sizeof(*id));
	unsigned char rnd[12];
	FILE* f = fopen("/dev/urandom", "rb");
	if (!f || fread(rnd, 1, sizeof(rnd), f) != sizeof(rnd)) { if (f) fclose(f); return ncclSystemError; }
	fclose(f);
	char* p = id->internal;
	p += snprintf(p, 16, "standin_");
	for (unsigned char b : rnd) p += snprintf(p, 3, "%02x", b);
	return ncclSuccess;
}

STANDIN_API ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
	if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
	id.internal[127] = 0;
	if (strncmp(id.internal, "standin_", 8) != 0) return ncclInvalidArgument;      // (an id that real RCCL made: the two libraries were mixed)
	ncclComm_t c = new ncclComm();
	c->dir = std::string("/dev/shm/sgp_rccl_") + id.internal;
	c->nranks = nranks; c->rank = rank;
	c->send_seq.assign(nranks, 0); c->recv_seq.assign(nranks, 0);
	if (mkdir(c->dir.c_str(), 0700) != 0 && errno != EEXIST) { delete c; return ncclSystemError; }
	// rendezvous: everybody announces itself and waits for everybody (as ncclCommInitRank blocks until all ranks have called it)
	char name[64]; snprintf(name, sizeof(name), "/hello_%d", rank);
	const int one = 1;
	if (!write_file(c->dir + name, &one, sizeof(one))) { delete c; return ncclSystemError; }
	for (int r = 0; r < nranks; ++r) { int v; snprintf(name, sizeof(name), "/hello_%d", r); if (!read_file(c, c->dir + name, &v, sizeof(v), false)) { delete c; return ncclSystemError; } }
	*comm = c;
	return ncclSuccess;
}

STANDIN_API ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { if (!comm || !count) return ncclInvalidArgument; *count = comm->nranks; return ncclSuccess; }

STANDIN_API ncclResult_t ncclCommAbort(ncclComm_t comm)
{
	if (!comm) return ncclInvalidArgument;
	comm->aborted = true;
	const int one = 1; write_file(comm->dir + "/abort", &one, sizeof(one));      // (peers waiting in a receive give up)
	return ncclSuccess;
}

STANDIN_API ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	if (!comm) return ncclInvalidArgument;
	// the last rank to leave removes the directory (best effort: a crashed peer leaves files behind in /dev/shm, named by the run's id)
	char name[64]; snprintf(name, sizeof(name), "/bye_%d", comm->rank);
	const int one = 1; write_file(comm->dir + name, &one, sizeof(one));
	int gone = 0;
	for (int r = 0; r < comm->nranks; ++r) { snprintf(name, sizeof(name), "/bye_%d", r); if (access((comm->dir + name).c_str(), F_OK) == 0) ++gone; }
	if (gone == comm->nranks) {
		if (DIR* d = opendir(comm->dir.c_str())) { while (dirent* e = readdir(d)) { if (e->d_name[0] != '.') unlink((comm->dir + "/" + e->d_name).c_str()); } closedir(d); }
		rmdir(comm->dir.c_str());
	}
	delete comm;
	return ncclSuccess;
}

STANDIN_API const char* ncclGetErrorString(ncclResult_t r)
{
	switch (r) {
	case ncclSuccess: return "no error";
	case ncclUnhandledCudaError: return "unhandled HIP error (test stand-in)";
	case ncclSystemError: return "system error: a message did not arrive, or its size is not what the receiver posted (test stand-in)";
	case ncclInvalidArgument: return "invalid argument (test stand-in)";
	default: return "error (test stand-in)";
	}
}

STANDIN_API ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
	if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
	const size_t bytes = sendcount * type_size(datatype);
	if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
	std::vector<char> mine(bytes), all(bytes * (size_t)comm->nranks);
	if (bytes && hipMemcpy(mine.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
	const uint64_t seq = comm->ag_seq++;
	char name[96]; snprintf(name, sizeof(name), "/ag_%llu_%d", (unsigned long long)seq, comm->rank);
	if (!write_file(comm->dir + name, mine.data(), bytes)) return ncclSystemError;
	for (int r = 0; r < comm->nranks; ++r) {
		snprintf(name, sizeof(name), "/ag_%llu_%d", (unsigned long long)seq, r);
		if (!read_file(comm, comm->dir + name, all.data() + bytes * (size_t)r, bytes, false)) return ncclSystemError;
	}
	// this rank's contribution of two gathers ago has been read by everybody (each of them has since contributed to the gather after it)
	if (seq >= 2) { snprintf(name, sizeof(name), "/ag_%llu_%d", (unsigned long long)(seq - 2), comm->rank); unlink((comm->dir + name).c_str()); }
	if (bytes && hipMemcpy(recvbuff, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
	return ncclSuccess;
}

STANDIN_API ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }
STANDIN_API ncclResult_t ncclGroupEnd()
{
	if (g_group_depth <= 0) return ncclInvalidUsage;
	if (--g_group_depth > 0) return ncclSuccess;
	std::vector<Op> ops; ops.swap(g_group_ops);
	return run_ops(ops);
}
STANDIN_API ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
	if (!comm || peer < 0 || peer >= comm->nranks || peer == comm->rank) return ncclInvalidArgument;
	Op op{ true, const_cast<void*>(sendbuff), count * type_size(datatype), peer, comm, stream };
	if (g_group_depth > 0) { g_group_ops.push_back(op); return ncclSuccess; }
	std::vector<Op> ops{ op };
	return run_ops(ops);
}
STANDIN_API ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
	if (!comm || peer < 0 || peer >= comm->nranks || peer == comm->rank) return ncclInvalidArgument;
	Op op{ false, recvbuff, count * type_size(datatype), peer, comm, stream };
	if (g_group_depth > 0) { g_group_ops.push_back(op); return ncclSuccess; }
	std::vector<Op> ops{ op };
	return run_ops(ops);
}
}
