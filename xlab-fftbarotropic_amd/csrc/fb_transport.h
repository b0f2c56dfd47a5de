// fb_transport.h -- the one collective of the multi-GPU path: an all-to-all of equal blocks
// between the ranks' exchange buffers, enqueued on a HIP stream.  Three implementations (fb_slab_comm.cpp):
//   rccl     grouped ncclSend/ncclRecv over xGMI (one process per GPU) -- the product transport
//   local    all ranks inside ONE process on ONE device (threads), device-to-device copies ordered by events:
//            the single-GPU rehearsal of the pipelined schedule (tests/test_gpu_slab.py)
//   callback the caller moves the bytes (e.g. torch.distributed/gloo between processes that share one GPU, MPI)
// No reference counterpart: the reference is single-process.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct fb_transport {
    void *self;
    int rank, world;
    // For every peer p (p == rank included): `count` floats at send + p*stride + offset of THIS rank arrive at
    // recv + rank*stride + offset of rank p.  Enqueued on `stream`; when that work has completed the receive
    // buffer is filled and the send buffer may be overwritten.  Every rank issues the same sequence of calls.
    int (*alltoall)(void *self, const float *send, float *recv, size_t stride, size_t offset, size_t count, hipStream_t stream);
    void (*destroy)(void *self);
    const char *name;
    // optional: what the transport's own communicator reports (RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice), -1 where unknown
    int (*info)(void *self, int *comm_ranks, int *comm_rank, int *device);
};

typedef int (*fb_alltoall_fn)(void *user, const float *send, float *recv, size_t stride, size_t offset, size_t count, void *hip_stream);

// fb_slab_comm.cpp
int fb_transport_rccl(fb_transport *tp, const char *unique_id, int rank, int world);
int fb_transport_local(fb_transport *tp, void *hub, int rank, int world);
int fb_transport_callback(fb_transport *tp, fb_alltoall_fn fn, void *user, int rank, int world);
