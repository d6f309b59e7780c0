// ingest_kernel.hip -- FPGA protocol-v2 datagrams -> float32 mic-major frame (the hot path's input layout).
//
// Reference: PC/src/receiver.c:94-151 (`receive_and_write_to_buffer`) / `receive_to_buffer`: one datagram per sample
// instant, `msg { u16 frequency; i8 n_arrays; i8 protocol_ver; i32 counter; i32 stream[N_MICROPHONES]; }`
// (PC/src/receiver.h:51-59); for array n, row y, column x the output mic index runs s = 0,1,2,... and reads
//     stream[n*ROWS*COLUMNS + y*COLUMNS + x]               on even rows,
//     stream[n*ROWS*COLUMNS + y*COLUMNS + COLUMNS - x]     on odd rows (serpentine wiring; note the reference's
//                                                          off-by-one: x = 0 reads the first element of the NEXT row),
// converts `(float)((double)v / NORM_FACTOR)` with NORM_FACTOR = 2^24, and writes data[s*N_SAMPLES + step].
// (float)((double)v / 2^24) == (float)v * 2^-24 exactly: int32 -> float rounds to nearest-even once, the scaling is a
// power of two -- so v_cvt_f32_i32 + v_mul_f32 reproduces it bit for bit.
// The kernel is a 64 x 64 transpose through LDS: datagram-major reads and mic-major writes are both coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bf {

namespace {

__global__ void __launch_bounds__(256) ingest_kernel(const unsigned char* __restrict__ packets, int packet_stride, int header_bytes,
                                                     int n_samples, int n_mics_out, int stream_len, int rows, int columns,
                                                     float scale, float* __restrict__ frame)
{
    __shared__ float tile[64][65];
    const int s0 = blockIdx.x * 64;      // first output mic of the tile
    const int t0 = blockIdx.y * 64;      // first sample of the tile
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int per = rows * columns;
    // read: thread tx -> mic s0+tx (coalesced within a datagram up to the serpentine permutation), 16 datagrams per pass
    {
        const int s = s0 + tx;
        int idx = -1;
        if (s < n_mics_out) {
            const int n = s / per, rem = s - n * per, y = rem / columns, x = rem - y * columns;
            const int row = n * per + y * columns;
            idx = (y & 1) ? row + columns - x : row + x;
            if (idx >= stream_len) idx = -1;     // the reference reads one int past the datagram here; we define it as 0
        }
        for (int r = ty; r < 64; r += 4) {
            const int t = t0 + r;
            float v = 0.0f;
            if (idx >= 0 && t < n_samples) {
                const int32_t raw = *reinterpret_cast<const int32_t*>(packets + (size_t)t * packet_stride + header_bytes + 4 * (size_t)idx);
                v = (float)raw * scale;
            }
            tile[r][tx] = v;
        }
    }
    __syncthreads();
    // write: thread tx -> sample t0+tx of mic s0+r (coalesced along the sample axis)
    for (int r = ty; r < 64; r += 4) {
        const int s = s0 + r, t = t0 + tx;
        if (s < n_mics_out && t < n_samples) frame[(size_t)s * n_samples + t] = tile[tx][r];
    }
}

}  // namespace

hipError_t launch_ingest(const void* d_packets, int packet_stride, int header_bytes, int n_samples, int n_mics_out, int stream_len,
                         int rows, int columns, float* d_frame, hipStream_t stream)
{
    const dim3 grid((unsigned)((n_mics_out + 63) / 64), (unsigned)((n_samples + 63) / 64));
    hipLaunchKernelGGL(ingest_kernel, grid, dim3(256), 0, stream, static_cast<const unsigned char*>(d_packets), packet_stride, header_bytes,
                       n_samples, n_mics_out, stream_len, rows, columns, 1.0f / 16777216.0f, d_frame);
    return hipGetLastError();
}

}  // namespace bf
