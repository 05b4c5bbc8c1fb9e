// C ABI of include/rdr_gather.h on RCCL: the gather of finished frames to rank 0 as one group of point-to-point
// transfers, and the all-gather of per-frame luminance. One communicator per process, one process per GPU.
//
// The reference has no multi-device path (one vk::Queue, Renderer/VulkanRenderer.cpp:548); this file has no
// counterpart there.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>

#include "../../include/rdr_gather.h"

struct rdr_gather {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = 0;
    std::string err;
};

namespace {

static_assert(sizeof(ncclUniqueId) == RDR_GATHER_ID_BYTES, "the rendezvous token is an ncclUniqueId");

int fail(rdr_gather* g, int code, const char* what, const char* detail) {
    if (g) g->err = std::string(what) + ": " + detail;
    return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(g, RDR_GATHER_ERR_DEVICE, #expr, hipGetErrorString(e_)); } while (0)
#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return fail(g, RDR_GATHER_ERR_DEVICE, #expr, ncclGetErrorString(r_)); } while (0)

}  // namespace

extern "C" {

int rdr_gather_make_id(uint8_t out_id[RDR_GATHER_ID_BYTES]) {
    if (!out_id) return RDR_GATHER_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return RDR_GATHER_ERR_DEVICE;
    memcpy(out_id, &id, RDR_GATHER_ID_BYTES);
    return RDR_GATHER_OK;
}

int rdr_gather_create(const uint8_t id_bytes[RDR_GATHER_ID_BYTES], int rank, int world, int device, rdr_gather** out) {
    if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world || device < 0) return RDR_GATHER_ERR_INVALID;
    rdr_gather* g = new (std::nothrow) rdr_gather();
    if (!g) return RDR_GATHER_ERR_DEVICE;
    g->rank = rank; g->world = world; g->device = device;
    *out = g;  // handed out even on failure, so that the caller can read the message; rdr_gather_destroy takes it either way
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    ncclUniqueId id;
    memcpy(&id, id_bytes, RDR_GATHER_ID_BYTES);
    NCCL_TRY(ncclCommInitRank(&g->comm, world, id, rank));
    return RDR_GATHER_OK;
}

void rdr_gather_destroy(rdr_gather* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->comm) (void)ncclCommDestroy(g->comm);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

const char* rdr_gather_last_error(const rdr_gather* g) { return g ? g->err.c_str() : "null communicator"; }

int rdr_gather_frames(rdr_gather* g, const void* send_device, void* recv_device, size_t bytes) {
    if (!g || !g->comm || !send_device || bytes == 0) return fail(g, RDR_GATHER_ERR_INVALID, "gather_frames", "null or empty buffer");
    if (g->rank == 0 && !recv_device) return fail(g, RDR_GATHER_ERR_INVALID, "gather_frames", "rank 0 needs a receive buffer of world * bytes");
    HIP_TRY(hipSetDevice(g->device));
    // one group: the send of this rank and, on rank 0, one receive per rank are posted together and progress
    // concurrently, each over the link to its peer
    NCCL_TRY(ncclGroupStart());
    ncclResult_t r = ncclSend(send_device, bytes, ncclUint8, 0, g->comm, g->stream);
    if (g->rank == 0)
        for (int src = 0; src < g->world && r == ncclSuccess; src++)
            r = ncclRecv((uint8_t*)recv_device + (size_t)src * bytes, bytes, ncclUint8, src, g->comm, g->stream);
    const ncclResult_t end = ncclGroupEnd();
    NCCL_TRY(r);
    NCCL_TRY(end);
    return RDR_GATHER_OK;
}

int rdr_gather_all_floats(rdr_gather* g, const float* mine_device, float* all_device, size_t count) {
    if (!g || !g->comm || !mine_device || !all_device || count == 0) return fail(g, RDR_GATHER_ERR_INVALID, "gather_all_floats", "null or empty buffer");
    HIP_TRY(hipSetDevice(g->device));
    NCCL_TRY(ncclAllGather(mine_device, all_device, count, ncclFloat, g->comm, g->stream));
    return RDR_GATHER_OK;
}

int rdr_gather_wait(rdr_gather* g) {
    if (!g) return RDR_GATHER_ERR_INVALID;
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (g->comm) {
        ncclResult_t async = ncclSuccess;
        NCCL_TRY(ncclCommGetAsyncError(g->comm, &async));
        NCCL_TRY(async);
    }
    return RDR_GATHER_OK;
}

}  // extern "C"
