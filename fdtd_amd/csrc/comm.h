// Thin wrapper over RCCL point-to-point calls for the slab halo exchange.
// RCCL is opened at run time (dlopen) the first time a communicator is asked
// for, so single-GPU use never loads it.
#pragma once

#include <cstddef>

namespace fdtd {

struct Comm;
struct LocalGroup;

int comm_unique_id(char id[128]);
int comm_create(const char id[128], int rank, int world, Comm **out);
// In-process transport: the ranks are contexts of ONE process (one host thread each, on one GPU or
// on several), a message is a device-to-device copy after a rendezvous of the two threads.  Same
// call sequence as the RCCL transport, so the exchange schedule can be exercised without a
// second process.
LocalGroup *local_group_create(int world);
void local_group_destroy(LocalGroup *g);
int comm_create_local(LocalGroup *g, int rank, Comm **out);
void comm_destroy(Comm *c);
int comm_group_start(Comm *c);
int comm_group_end(Comm *c);
// count floats to / from `peer`, queued on `stream` (a hipStream_t)
int comm_send(Comm *c, const float *buf, size_t count, int peer, void *stream);
int comm_recv(Comm *c, float *buf, size_t count, int peer, void *stream);

} // namespace fdtd
