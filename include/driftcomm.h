/* driftcomm.h — C ABI of libdriftcomm: the two collectives of the per-m hot path over RCCL (xGMI), for hosts that
 * drive several GPUs WITHOUT torch.distributed (the Python layer of this repository uses torch.distributed's "nccl"
 * backend — which is RCCL — through driftscan_amd/parallel.py and does not load this library).
 *
 * m-blocks are independent: there is no data-path collective.  What the reference exchanges is
 *   - the spectra gathered to rank 0:   caput.mpiutil gathers inside collect_m_array, drift/core/kltransform.py:21-52,
 *                                        beamtransfer.py:931-947                                   -> dm_gather_f64
 *   - the Fisher matrix / bias summed:  mpiutil.world.Allreduce, drift/core/psestimation.py:506-507 -> dm_allreduce_f64
 * One communicator per process = one rank per GPU.  The 128-byte id made by dm_comm_unique_id on rank 0 travels to the
 * other ranks by whatever channel the host has (a file, a socket, MPI_Bcast).
 * Every function returns 0 on success, < 0 on failure (dm_comm_last_error()).
 *
 * Stream ordering.  A collective is enqueued on the communicator's stream (the one given to dm_comm_init_rank, or a
 * private non-blocking stream).  Its buffers are produced and consumed on the CALLER's stream, which every collective
 * takes as `user_stream` (a hipStream_t; NULL = the legacy default stream): the library makes the communicator's stream
 * wait for everything enqueued on `user_stream` so far before the collective starts, and makes `user_stream` wait for the
 * collective afterwards (events, nothing blocks the host).  So kernels enqueued on `user_stream` before the call have
 * finished writing the buffer when it is read, and kernels enqueued on it after the call see the result.  Work on any
 * OTHER stream is the caller's to order; the host itself reads a result only after dm_comm_sync (or a synchronisation
 * of `user_stream`).
 */
#ifndef DRIFTCOMM_H
#define DRIFTCOMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dm_comm dm_comm;

#define DM_COMM_ID_BYTES 128

const char* dm_comm_last_error(void);
/* rank 0: fill id_out[DM_COMM_ID_BYTES] */
int dm_comm_unique_id(void* id_out);
/* every rank (collective): device = the GPU this rank drives; stream = hipStream_t the collectives are enqueued on, or NULL */
int dm_comm_init_rank(int nranks, int rank, const void* id, int device, void* stream, dm_comm** out);
int dm_comm_destroy(dm_comm* comm);
int dm_comm_rank(const dm_comm* comm);
int dm_comm_size(const dm_comm* comm);
/* in place, SUM over ranks, n doubles of device memory; ordered against user_stream as described above */
int dm_allreduce_f64(dm_comm* comm, double* data_dev, size_t n, void* user_stream);
/* every rank sends n doubles; rank `root` receives nranks * n (rank-major) in recv_dev (ignored elsewhere) */
int dm_gather_f64(dm_comm* comm, const double* send_dev, double* recv_dev, size_t n, int root, void* user_stream);
/* block the host until everything enqueued so far on the communicator's stream is done */
int dm_comm_sync(dm_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* DRIFTCOMM_H */
