// C ABI of the native RCCL hop (rccl_edge.cpp), shared with the native stage loop.
#pragma once
#include <cstddef>
#include <cstdint>

// "no stream" for the after_stream arguments: a null handle is the default stream, not absence
#define DLIA_NO_STREAM ((void*)(intptr_t)-1)

extern "C" {
const char* dlia_comm_last_error();
int dlia_comm_init(const char* rccl_path);
int dlia_comm_version();
int dlia_comm_unique_id(void* out128);
int dlia_edge_open(const void* uid128, int nranks, int rank, int peer, int device, void** out);
int dlia_edge_send(void* edge, const void* ptr, size_t nbytes, void* after_stream);
int dlia_edge_recv(void* edge, void* ptr, size_t nbytes, void* after_stream, void* after_event);
int dlia_edge_sendrecv(void* edge, const void* sptr, void* rptr, size_t nbytes, void* after_stream);
int dlia_edge_send_on(void* edge, const void* ptr, size_t nbytes, void* stream);
int dlia_edge_recv_on(void* edge, void* ptr, size_t nbytes, void* stream);
int dlia_edge_sendrecv_on(void* edge, const void* sptr, void* rptr, size_t nbytes, void* stream);
int dlia_edge_wait(void* edge, int token, void* stream);
int dlia_edge_query(void* edge, int token);
int dlia_edge_sync(void* edge, int token, int timeout_ms);
void* dlia_edge_stream(void* edge);
void dlia_edge_stats(void* edge, uint64_t* out4);
int dlia_edge_abort(void* edge);
int dlia_edge_close(void* edge);
}
