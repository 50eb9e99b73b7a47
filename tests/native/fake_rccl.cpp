// Test infrastructure: a stand-in for the RCCL entry points lbmhip_halo_exchange_rccl looks up in the process, so that the
// send/recv call order of several "ranks" can be checked on ONE GPU.  A communicator is a pointer to the rank's number.
// Calls are recorded; fake_rccl_deliver() then moves the data the way RCCL matches point-to-point messages: the k-th send
// from rank a to rank b meets the k-th receive rank b posted for rank a.  Never loaded by the product.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <vector>

namespace {
struct Message { int rank, peer; void* buffer; size_t bytes; bool used; };
std::vector<Message> g_sends, g_recvs;
int g_depth = 0, g_outside_group = 0;
}

extern "C" {
int ncclGroupStart(void) { g_depth++; return 0; }
int ncclGroupEnd(void) { g_depth--; return g_depth < 0 ? 1 : 0; }
const char* ncclGetErrorString(int) { return "fake RCCL error"; }
int ncclSend(void* buffer, size_t count, int datatype, int peer, void* comm, hipStream_t) {
	if (datatype != 1 || !comm) return 1; // the library sends bytes (ncclUint8)
	if (g_depth <= 0) g_outside_group++;
	g_sends.push_back({*static_cast<int*>(comm), peer, buffer, count, false});
	return 0;
}
int ncclRecv(void* buffer, size_t count, int datatype, int peer, void* comm, hipStream_t) {
	if (datatype != 1 || !comm) return 1;
	if (g_depth <= 0) g_outside_group++;
	g_recvs.push_back({*static_cast<int*>(comm), peer, buffer, count, false});
	return 0;
}
// returns the number of messages delivered, or a negative number: -1 unmatched receive, -2 size mismatch, -3 send left
// over, -4 a call outside a group, -5 HIP error
int fake_rccl_deliver(void) {
	if (hipDeviceSynchronize() != hipSuccess) return -5;
	int delivered = 0, status = 0;
	for (Message& r : g_recvs) {
		Message* match = nullptr;
		for (Message& s : g_sends) if (!s.used && s.rank == r.peer && s.peer == r.rank) { match = &s; break; }
		if (!match) { status = -1; break; }
		if (match->bytes != r.bytes) { status = -2; break; }
		if (hipMemcpy(r.buffer, match->buffer, r.bytes, hipMemcpyDeviceToDevice) != hipSuccess) { status = -5; break; }
		match->used = true;
		delivered++;
	}
	if (status == 0) for (const Message& s : g_sends) if (!s.used) status = -3;
	if (status == 0 && (g_outside_group || g_depth != 0)) status = -4;
	g_sends.clear(); g_recvs.clear(); g_outside_group = 0;
	return status ? status : delivered;
}
}
