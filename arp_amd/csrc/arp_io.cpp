// Host-side I/O of the labelling path: parallel inflate of the stored chunks of a gzip-chunked
// HDF5 dataset, keeping only the frames the labelling pass consumes.
//
// The demonstration file (reference writer: data/PPG/trajectory_recorder.py:148-176) stores `ob` as uint8
// [len, num_frames, H, W, 3] in gzip chunks of ONE row; the reference reads g[img_key][traj, -1] (arp_dt/label_reward.py:268), i.e.
// inflates every row's 8-frame chunk for one frame.  arp_amd/h5store.py asks libhdf5 where the chunks it needs are stored
// (H5Dget_chunk_info_by_coord) and hands the (address, size) list to this function, which pread()s and inflates them on native
// threads -- no Python, no library lock on the data path -- and copies the LAST cnt[i] frames of chunk i to their place in the
// output (a row's chunk holds the last frames of num_frames consecutive rows: trajectory_recorder.py:103-115).
#include <zlib.h>

#include <atomic>
#include <cerrno>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/arp_hip.h"

namespace arp {
void set_error(const std::string& msg);
int fail(const std::string& msg);
}  // namespace arp

extern "C" int arp_h5_inflate_last_frames(int fd, int n, const uint64_t* addr, const uint64_t* size, const uint8_t* stored_raw,
                                          uint64_t chunk_bytes, uint64_t frame_bytes, const uint64_t* dst_off, const uint32_t* cnt,
                                          uint8_t* out, int threads) {
    using arp::fail;
    if (n < 0 || (n > 0 && (!addr || !size || !dst_off || !cnt || !out)) || frame_bytes == 0 || chunk_bytes % frame_bytes)
        return fail("arp_h5_inflate_last_frames: bad argument");
    if (n == 0) return 0;
    const uint64_t frames_per_chunk = chunk_bytes / frame_bytes;
    for (int i = 0; i < n; ++i)
        if (cnt[i] == 0 || cnt[i] > frames_per_chunk) return fail("arp_h5_inflate_last_frames: cnt out of range");
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads > n) threads = n;
    if (threads < 1) threads = 1;

    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    std::mutex err_mu;
    std::string err;
    auto report = [&](const std::string& m) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!failed.exchange(true)) err = m;
    };
    auto work = [&]() {
        std::vector<uint8_t> stored, chunk(chunk_bytes);
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            uint8_t* dst = out + dst_off[i];
            const uint64_t keep = (uint64_t)cnt[i] * frame_bytes;
            if (size[i] == 0) {  // chunk never written: the dataset's fill value (0)
                memset(dst, 0, keep);
                continue;
            }
            stored.resize(size[i]);
            uint64_t got = 0;
            while (got < size[i]) {
                const ssize_t r = pread(fd, stored.data() + got, size[i] - got, (off_t)(addr[i] + got));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) {
                    report("pread of chunk " + std::to_string(i) + " failed: " + (r < 0 ? strerror(errno) : "end of file"));
                    return;
                }
                got += (uint64_t)r;
            }
            if (stored_raw && stored_raw[i]) {  // the deflate filter was skipped for this chunk (filter mask)
                if (size[i] != chunk_bytes) {
                    report("unfiltered chunk " + std::to_string(i) + " has the wrong size");
                    return;
                }
                memcpy(dst, stored.data() + (chunk_bytes - keep), keep);
                continue;
            }
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit(&zs) != Z_OK) {
                report("inflateInit failed");
                return;
            }
            zs.next_in = stored.data();
            zs.avail_in = (uInt)size[i];
            zs.next_out = chunk.data();
            zs.avail_out = (uInt)chunk_bytes;
            const int rc = inflate(&zs, Z_FINISH);
            const uint64_t produced = zs.total_out;
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || produced != chunk_bytes) {
                report("chunk " + std::to_string(i) + " did not inflate to " + std::to_string(chunk_bytes) + " bytes (zlib rc " +
                       std::to_string(rc) + ", " + std::to_string(produced) + " bytes)");
                return;
            }
            memcpy(dst, chunk.data() + (chunk_bytes - keep), keep);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (failed.load()) return fail("arp_h5_inflate_last_frames: " + err);
    return 0;
}

// ---- writing the reward datasets (label_reward.py:273-289): gzip chunks of ONE row of num_frames float32 (32 bytes) ----------------
// Through H5Dwrite the library runs its filter pipeline per chunk -- a deflateInit / deflateEnd pair and a 256 KB state allocation for
// every 32-byte row -- 39 ms per 8192-row dataset on the MI355X host.  Here every row is deflated with ONE reused z_stream and handed
// to H5Dwrite_chunk (the caller passes the function's address inside the libhdf5 it has loaded, so this file needs no HDF5 headers):
// the library only places the chunk.  The stream is the zlib format at `level`, i.e. what the deflate filter itself produces
// (compress2), so any HDF5 reader inflates it.
extern "C" int arp_h5_write_rows_deflated(void* write_chunk_fn, int64_t dset, int64_t dxpl, const void* rows, uint64_t row_bytes, uint64_t n_rows,
                                          uint64_t first_row, int ndim, int level) {
    using arp::fail;
    typedef int (*write_chunk_t)(int64_t, int64_t, uint32_t, const unsigned long long*, size_t, const void*);
    if (!write_chunk_fn || (!rows && n_rows) || row_bytes == 0 || ndim < 1 || ndim > 8 || level < 0 || level > 9)
        return fail("arp_h5_write_rows_deflated: bad argument");
    write_chunk_t wc = reinterpret_cast<write_chunk_t>(write_chunk_fn);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit(&zs, level) != Z_OK) return fail("arp_h5_write_rows_deflated: deflateInit failed");
    std::vector<uint8_t> buf(deflateBound(&zs, (uLong)row_bytes) + 16);
    unsigned long long off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc = 0;
    const uint8_t* src = static_cast<const uint8_t*>(rows);
    for (uint64_t i = 0; i < n_rows && rc == 0; ++i) {
        if (deflateReset(&zs) != Z_OK) { rc = fail("arp_h5_write_rows_deflated: deflateReset failed"); break; }
        zs.next_in = const_cast<Bytef*>(src + i * row_bytes);
        zs.avail_in = (uInt)row_bytes;
        zs.next_out = buf.data();
        zs.avail_out = (uInt)buf.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { rc = fail("arp_h5_write_rows_deflated: deflate failed"); break; }
        off[0] = first_row + i;
        if (wc(dset, dxpl, 0u, off, (size_t)zs.total_out, buf.data()) < 0) rc = fail("H5Dwrite_chunk failed at row " + std::to_string(first_row + i));
    }
    deflateEnd(&zs);
    return rc;
}
