/*
 * rdr_gather.h — C ABI of the multi-GPU exchange step for hosts that drive include/rdr.h directly (C, C++).
 *
 * No counterpart in the reference, which renders on one device through one queue (Renderer/VulkanRenderer.cpp:548)
 * and presents to a window: a main.cpp-style host (Renderer/main.cpp:432-462) that renders a frame batch on several
 * GPUs, one process and one rdr_ctx per GPU with frame k on rank k mod R, uses these calls to bring the finished
 * RGBA8 frames to rank 0 and, for the auto-exposure recurrence over a sharded sequence
 * (VulkanRenderer.cpp:1075-1082), to exchange one average log-luminance per frame.
 *
 * The library (renderer_amd/lib/librdr_gather.so) is separate from librdr_hip.so so that single-GPU hosts do not load
 * RCCL. Transport: RCCL over xGMI. The gather is a group of point-to-point transfers (every rank sends to rank 0,
 * rank 0 posts one receive per rank): rank 0 takes in over one xGMI link per peer at once, where a ring would
 * pass every frame over a single link.
 *
 * Conventions as in rdr.h: plain C types, int status (0 = RDR_GATHER_OK), message from rdr_gather_last_error.
 * A communicator is externally synchronised and bound to one GPU.
 */
#ifndef RDR_GATHER_H
#define RDR_GATHER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDR_GATHER_OK 0
#define RDR_GATHER_ERR_INVALID 1
#define RDR_GATHER_ERR_DEVICE 2   /* HIP or RCCL error; the message has the library's text */

#define RDR_GATHER_ID_BYTES 128   /* size of the rendezvous token (an ncclUniqueId) */

typedef struct rdr_gather rdr_gather;

/* Rank 0 makes the rendezvous token and hands it to the other ranks by whatever channel the launcher offers (a file,
 * an environment variable, a socket); every rank then passes the same bytes to rdr_gather_create. */
int rdr_gather_make_id(uint8_t out_id[RDR_GATHER_ID_BYTES]);

/* Joins the communicator of `world` ranks as rank `rank` on HIP device `device` (collective: returns once every rank
 * has called it). Call it from a process that renders on that device with its own rdr_ctx. */
int rdr_gather_create(const uint8_t id[RDR_GATHER_ID_BYTES], int rank, int world, int device, rdr_gather** out);
void rdr_gather_destroy(rdr_gather* g);
const char* rdr_gather_last_error(const rdr_gather* g);

/* Every rank sends `bytes` bytes at `send_device` to rank 0; on rank 0 `recv_device` (world * bytes, device memory)
 * receives rank r's block at offset r * bytes; other ranks pass NULL. The transfers are enqueued on the
 * communicator's own stream and the call returns; rdr_gather_wait waits for them. The caller makes sure the frame
 * in `send_device` is complete first (rdr_wait_frame on the rdr_ctx that rendered it) and leaves both buffers alone
 * until rdr_gather_wait has returned: the usual loop enqueues frame k + 1, waits for frame k, starts its gather and
 * goes on to frame k + 2, so that the gather of a frame overlaps the rendering of the next. */
int rdr_gather_frames(rdr_gather* g, const void* send_device, void* recv_device, size_t bytes);

/* All ranks receive `count` floats from every rank: all_device[r * count + i] = rank r's mine_device[i] (device
 * memory on both sides). Same stream, same completion rule as rdr_gather_frames. */
int rdr_gather_all_floats(rdr_gather* g, const float* mine_device, float* all_device, size_t count);

/* Waits until everything this communicator has enqueued so far has finished on this rank. */
int rdr_gather_wait(rdr_gather* g);

#ifdef __cplusplus
}
#endif
#endif /* RDR_GATHER_H */
