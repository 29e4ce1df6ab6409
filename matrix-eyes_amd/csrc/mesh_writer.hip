// output.rs:195-261 output_mesh and its two writers (ObjWriter :484-630, PlyWriter :385-482).
// Vertex ids, faces and coordinates come from the GPU kernels in output.hip; this file is the
// host-side serialisation, byte-for-byte the reference's text/binary layout.
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "model.h"
#include "ryu_f64.h"

using namespace me;

namespace {

// Rust `{}` for f64: shortest digits that round-trip, positional notation, "1" for 1.0, "-0" for -0.0, "NaN" / "inf" /
// "-inf" (ryu_f64.h: the formatter the device kernels of obj_format.hip run; std::to_chars(fixed) gives the same
// bytes below 2^53 and prints exact integer digits beyond, where Rust pads the shortest digits with zeros).
inline void put_f64(std::string& out, double v) {
    char buf[ryu::kMaxFixedChars + 8];
    out.append(buf, (size_t)ryu::format_fixed(v, buf));
}

inline void put_u64(std::string& out, unsigned long long v) {
    char buf[24];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);
    out.append(buf, r.ptr);
}

struct FileSink {
    FILE* f = nullptr;
    std::string buf;
    explicit FileSink(const std::string& path) {
        f = fopen(path.c_str(), "wb");
        ME_CHECK(f, ME_ERR_IO, "cannot create %s: %s", path.c_str(), strerror(errno));
        buf.reserve(1 << 20);
    }
    ~FileSink() {
        if (f) fclose(f);
    }
    void flush_if_full() {  // WRITE_BUFFER_SIZE, output.rs:383
        if (buf.size() >= (1u << 20)) flush();
    }
    void flush() {
        if (!buf.empty()) {
            ME_CHECK(fwrite(buf.data(), 1, buf.size(), f) == buf.size(), ME_ERR_IO, "write failed: %s",
                     strerror(errno));
            buf.clear();
        }
    }
    void close() {
        flush();
        const int r = fclose(f);
        f = nullptr;
        ME_CHECK(r == 0, ME_ERR_IO, "close failed: %s", strerror(errno));
    }
};

// One section of the file (all "vt" lines, all "v" lines, all faces ...): `item(i, out)` appends item i.
// A 1536^2 textured OBJ is 450 MB of shortest-round-trip decimals, 0.59 s single-threaded beside a 26 ms
// forward pass; the items are independent, so chunks of 32 Ki items are formatted by up to 12 host threads
// into their own strings and copied into the file with pwrite at the offsets the sizes give (0.23 s): the
// bytes are those of the sequential loop.
template <typename Item>
void write_section(FileSink& w, int64_t count, Item item, bool threads = true) {
    constexpr int64_t kChunk = 32768;
    const int64_t nchunks = (count + kChunk - 1) / kChunk;
    unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = threads ? (int)std::min<int64_t>(nchunks, hw ? (hw > 12 ? 12 : hw) : 4) : 1;
    if (nthreads <= 1) {
        for (int64_t i = 0; i < count; ++i) {
            w.flush_if_full();
            item(i, w.buf);
        }
        return;
    }
    w.flush();
    std::vector<std::string> chunks((size_t)nchunks);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (int64_t c; (c = next.fetch_add(1)) < nchunks;) {
            std::string& out = chunks[(size_t)c];
            out.reserve(1 << 20);
            const int64_t end = std::min(count, (c + 1) * kChunk);
            for (int64_t i = c * kChunk; i < end; ++i) item(i, out);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
    // the same threads then copy their chunks into the file at the offsets the sizes give
    ME_CHECK(fflush(w.f) == 0, ME_ERR_IO, "write failed: %s", strerror(errno));
    const off_t base = ftello(w.f);
    std::vector<off_t> offset((size_t)nchunks + 1, base);
    for (int64_t c = 0; c < nchunks; ++c) offset[(size_t)c + 1] = offset[(size_t)c] + (off_t)chunks[(size_t)c].size();
    const int fd = fileno(w.f);
    std::atomic<int64_t> next_w{0};
    std::atomic<int> failed{0};
    auto put = [&]() {
        for (int64_t c; (c = next_w.fetch_add(1)) < nchunks;) {
            const std::string& out = chunks[(size_t)c];
            size_t done = 0;
            while (done < out.size()) {
                const ssize_t r = pwrite(fd, out.data() + done, out.size() - done, offset[(size_t)c] + (off_t)done);
                if (r <= 0) {
                    failed = errno ? errno : EIO;
                    return;
                }
                done += (size_t)r;
            }
            std::string().swap(chunks[(size_t)c]);
        }
    };
    pool.clear();
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(put);
    put();
    for (std::thread& t : pool) t.join();
    ME_CHECK(failed == 0, ME_ERR_IO, "write failed: %s", strerror(failed));
    ME_CHECK(fseeko(w.f, offset[(size_t)nchunks], SEEK_SET) == 0, ME_ERR_IO, "seek failed: %s", strerror(errno));
}

bool ends_with_ci(const std::string& s, const char* suffix) {
    const size_t n = strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
        if (tolower((unsigned char)s[s.size() - n + i]) != suffix[i]) return false;
    return true;
}

// Path::file_stem / Path::parent for '/'-separated paths
std::string file_stem(const std::string& path) {
    const size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot != 0) name = name.substr(0, dot);
    return name;
}
std::string parent_dir(const std::string& path) {
    const size_t slash = path.find_last_of('/');
    if (slash == std::string::npos) return "";
    return slash == 0 ? "/" : path.substr(0, slash);
}

void put_be64(std::string& out, double v) {
    uint64_t u;
    memcpy(&u, &v, 8);
    for (int i = 7; i >= 0; --i) out.push_back((char)((u >> (8 * i)) & 0xff));
}
void put_be32(std::string& out, uint32_t u) {
    for (int i = 3; i >= 0; --i) out.push_back((char)((u >> (8 * i)) & 0xff));
}

}  // namespace

namespace {

// Everything of output.rs:195-261 that is arithmetic, on the device: IndexedMesh::new + remap_face, the vertex
// coordinates, and (OBJ) the text itself.
struct DeviceMesh {
    int64_t nverts = 0, nfaces = 0;
    int32_t* vindex = nullptr;
    int32_t* faces = nullptr;
    float *uv = nullptr, *xyz = nullptr;
};

DeviceMesh build_mesh(me_ctx* ctx, const float* depth, int32_t width, int32_t height, uint32_t original_width,
                      uint32_t original_height) {
    DeviceMesh m;
    const size_t nv = (size_t)width * height;
    const size_t nt = 2 * (size_t)(width - 1) * (height - 1);
    const float* d = (const float*)to_device(ctx, depth, nv * 4, "out.depth");
    m.vindex = (int32_t*)site_buf(ctx, "out.vindex", nv * 4);
    m.faces = (int32_t*)site_buf(ctx, "out.faces", nt * 12);
    mesh_index_run(d, width, height, m.vindex, m.faces, &m.nverts, &m.nfaces,
                   site_buf(ctx, "out.mesh.ws", mesh_workspace_bytes(width, height)), ctx->stream);
    m.uv = (float*)site_buf(ctx, "out.uv", m.nverts * 8 + 8);
    m.xyz = (float*)site_buf(ctx, "out.xyz", m.nverts * 12 + 12);
    const uint32_t mx = original_width > original_height ? original_width : original_height;
    mesh_vertices_launch(d, width, height, m.vindex, (float)original_width / (float)mx,
                         (float)original_height / (float)mx, m.uv, m.xyz, ctx->stream);
    return m;
}

// The OBJ text in device memory (site buffer "out.objtext"): header lines + vt / v / f sections.
struct DeviceText {
    char* dev = nullptr;
    int64_t bytes = 0;
};
DeviceText obj_text_on_device(me_ctx* ctx, const DeviceMesh& m, int32_t width, int32_t height, const std::string& stem,
                              int32_t vertex_mode, const uint8_t* vertex_colors) {
    const bool tex = vertex_mode == ME_VERTEX_TEXTURE;
    const bool with_color = vertex_mode == ME_VERTEX_COLOR && vertex_colors;
    const uint8_t* vrgb = nullptr;
    if (with_color) {
        const size_t nv = (size_t)width * height;
        const uint8_t* pix = (const uint8_t*)to_device(ctx, vertex_colors, nv * 3, "out.pixel_rgb");
        uint8_t* v = (uint8_t*)site_buf(ctx, "out.vertex_rgb", (size_t)m.nverts * 3 + 16);
        obj_vertex_colors_launch(m.vindex, pix, (int64_t)nv, v, ctx->stream);
        vrgb = v;
    }
    std::string header;
    if (tex) header = "mtllib " + stem + ".mtl\nusemtl Textured\n";  // output.rs:556-562
    void* ws = site_buf(ctx, "out.objfmt.ws", obj_format_workspace_bytes(m.nverts, m.nfaces));
    DeviceText t;
    t.bytes = obj_format_measure(m.uv, m.xyz, vrgb, m.faces, m.nverts, m.nfaces, tex, (int64_t)header.size(), ws,
                                 ctx->stream);
    // the size follows the kept faces of each image: grown in 64 MiB steps so that a sequence of images settles
    const size_t step = (size_t)64 << 20;
    t.dev = (char*)site_buf(ctx, "out.objtext", ((size_t)t.bytes + 64 + step - 1) / step * step);
    if (!header.empty())
        ME_HIP(hipMemcpyAsync(t.dev, header.data(), header.size(), hipMemcpyHostToDevice, ctx->stream));
    obj_format_write(m.uv, m.xyz, vrgb, m.faces, m.nverts, m.nfaces, tex, t.dev, ws, ctx->stream);
    return t;
}

// `bytes` from pinned host memory into a new file: chunks of 8 MiB by up to 8 threads (page-cache copies scale
// with threads; one thread moves ~2 GB/s)
void write_file_parallel(const std::string& path, const char* data, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    ME_CHECK(f, ME_ERR_IO, "cannot create %s: %s", path.c_str(), strerror(errno));
    const int fd = fileno(f);
    constexpr size_t kChunk = 8u << 20;
    const int64_t nchunks = (int64_t)((bytes + kChunk - 1) / kChunk);
    // Two threads: the copy into the page cache runs at 2.4 - 2.7 GB/s per FILE with 2 or with 8 writers, and with one
    // rank per GPU writing at once eight writers per file cost the node half of what it can take (tools/
    // node_write_ceiling.py on the 256-core host: 8 processes x 2 threads 20.6 GB/s, x 8 threads 9.7 GB/s).  ME_WRITE_THREADS.
    static const int want = getenv("ME_WRITE_THREADS") ? std::max(1, std::min(64, atoi(getenv("ME_WRITE_THREADS")))) : 2;
    const int nthreads = (int)std::min<int64_t>(nchunks, want);
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    auto put = [&]() {
        for (int64_t c; (c = next.fetch_add(1)) < nchunks;) {
            size_t done = (size_t)c * kChunk;
            const size_t end = std::min(bytes, done + kChunk);
            while (done < end) {
                const ssize_t r = pwrite(fd, data + done, end - done, (off_t)done);
                if (r <= 0) {
                    failed = errno ? errno : EIO;
                    return;
                }
                done += (size_t)r;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(put);
    put();
    for (std::thread& t : pool) t.join();
    const int r = fclose(f);
    ME_CHECK(failed == 0, ME_ERR_IO, "write failed: %s", strerror(failed));
    ME_CHECK(r == 0, ME_ERR_IO, "close failed: %s", strerror(errno));
}

// pinned staging buffer of write slot `which` (the D2H copy of a few hundred MB runs at the link rate only from pinned
// memory)
char* pinned_buf(me_ctx* ctx, size_t bytes, int which = 0) {
    me_ctx::WriteSlot& w = ctx->write_slots[(size_t)which];
    if (w.pinned && w.pinned_bytes >= bytes) return (char*)w.pinned;
    if (w.pinned) ME_HIP(hipHostFree(w.pinned));
    w.pinned = nullptr, w.pinned_bytes = 0;
    const size_t step = (size_t)64 << 20;
    const size_t want = (bytes + step - 1) / step * step;
    ME_HIP(hipHostMalloc(&w.pinned, want, hipHostMallocDefault));
    w.pinned_bytes = want;
    return (char*)w.pinned;
}

void write_mtl(const std::string& dest, const std::string& stem, const char* source_path) {  // output.rs:525-547
    const std::string dir = parent_dir(dest);
    FileSink m((dir.empty() ? std::string() : dir + "/") + stem + ".mtl");
    m.buf += "newmtl Textured\nKa 0.2 0.2 0.2\nKd 0.8 0.8 0.8\nKs 1.0 1.0 1.0\nillum 2\n";
    m.buf += "Ns 0.000500\n";
    m.buf += std::string("map_Ka ") + source_path + "\n";
    m.buf += std::string("map_Kd ") + source_path + "\n\n";
    m.close();
}

}  // namespace

// waits for pending write k; a failure becomes the Error of the caller that waited for it
static void join_pending_write(me_ctx* ctx, int k) {
    me_ctx::WriteSlot& w = ctx->write_slots[(size_t)k];
    if (!w.active) return;
    if (w.th.joinable()) w.th.join();
    w.active = false;
    if (w.code) {
        const int32_t code = w.code;
        const std::string msg = w.msg;
        w.code = 0, w.msg.clear();
        fail(code, "write-behind: %s", msg.c_str());
    }
}

extern "C" int32_t me_ctx_set_write_behind(me_ctx* ctx, int32_t files_in_flight) {
    if (!ctx || files_in_flight < 0 || files_in_flight > 64) return ME_ERR_BAD_ARG;
    const int32_t rc = me_output_flush(ctx);  // nothing in flight while the slots change
    const int n = files_in_flight == 0 ? 0 : (files_in_flight < 2 ? 2 : files_in_flight);
    (void)hipSetDevice(ctx->device);
    for (size_t k = (size_t)(n > 1 ? n : 1); k < ctx->write_slots.size(); ++k)
        if (ctx->write_slots[k].pinned) (void)hipHostFree(ctx->write_slots[k].pinned);
    ctx->write_slots.resize((size_t)(n > 1 ? n : 1));
    ctx->write_behind = n;
    ctx->write_next = 0;
    return rc;
}

extern "C" int32_t me_output_flush(me_ctx* ctx) {
    if (!ctx) return ME_ERR_BAD_ARG;
    int32_t rc = ME_OK;
    // (output overlap: the output stream's queued kernels and copies too)
    if (ctx->output_overlap && ctx->out_stream && hipStreamSynchronize(ctx->out_stream) != hipSuccess)
        ctx->last_error = "me_output_flush: the output stream failed", rc = ME_ERR_HIP;
    for (int k = 0; k < (int)ctx->write_slots.size(); ++k) {
        try {
            join_pending_write(ctx, k);
        } catch (const me::Error& e) {
            if (rc == ME_OK) ctx->last_error = e.msg, rc = e.code;
        }
    }
    return rc;
}

extern "C" int32_t me_last_mesh_timing(const me_ctx* ctx, double ms_out[4], int64_t* text_bytes) {
    if (!ctx || !ms_out) return ME_ERR_BAD_ARG;
    for (int i = 0; i < 4; ++i) ms_out[i] = ctx->mesh_ms[i];
    if (text_bytes) *text_bytes = ctx->mesh_bytes;
    return ME_OK;
}

extern "C" int32_t me_mesh_obj_text(me_ctx* ctx, const float* depth, int32_t width, int32_t height,
                                    uint32_t original_width, uint32_t original_height, const char* stem,
                                    int32_t vertex_mode, const uint8_t* vertex_colors, const uint8_t** text_dev,
                                    int64_t* nbytes) {
    if (!ctx) return ME_ERR_BAD_ARG;
    try {
        ME_HIP(hipSetDevice(ctx->device));
        ME_CHECK(depth && stem && text_dev && nbytes, ME_ERR_BAD_ARG, "me_mesh_obj_text: null pointer");
        ME_CHECK(vertex_mode >= ME_VERTEX_PLAIN && vertex_mode <= ME_VERTEX_TEXTURE, ME_ERR_BAD_ARG,
                 "vertex mode %d", vertex_mode);
        ME_CHECK(width >= 2 && height >= 2, ME_ERR_BAD_SHAPE, "me_mesh_obj_text: %dx%d", width, height);
        ME_CHECK(original_width > 0 && original_height > 0, ME_ERR_BAD_ARG, "original size 0");
        OutputScope out_scope(ctx, depth);
        const DeviceMesh m = build_mesh(ctx, depth, width, height, original_width, original_height);
        const DeviceText t = obj_text_on_device(ctx, m, width, height, stem, vertex_mode, vertex_colors);
        ME_HIP(hipStreamSynchronize(ctx->stream));
        *text_dev = (const uint8_t*)t.dev, *nbytes = t.bytes;
    } catch (const me::Error& e) {
        ctx->last_error = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        ctx->last_error = std::string("internal: ") + e.what();
        return ME_ERR_BAD_ARG;
    }
    return ME_OK;
}

extern "C" int32_t me_output_mesh(me_ctx* ctx, const float* depth, int32_t width, int32_t height,
                                  uint32_t original_width, uint32_t original_height,
                                  const char* destination_path, const char* source_path,
                                  int32_t vertex_mode, const uint8_t* vertex_colors) {
    if (!ctx) return ME_ERR_BAD_ARG;
    try {
        ME_HIP(hipSetDevice(ctx->device));
        ME_CHECK(depth && destination_path && source_path, ME_ERR_BAD_ARG,
                 "me_output_mesh: null pointer");
        ME_CHECK(vertex_mode >= ME_VERTEX_PLAIN && vertex_mode <= ME_VERTEX_TEXTURE, ME_ERR_BAD_ARG,
                 "vertex mode %d", vertex_mode);
        ME_CHECK(width >= 2 && height >= 2, ME_ERR_BAD_SHAPE, "me_output_mesh: %dx%d", width, height);
        ME_CHECK(original_width > 0 && original_height > 0, ME_ERR_BAD_ARG, "original size 0");
        const std::string dest(destination_path);
        const bool ply = ends_with_ci(dest, ".ply"), obj = ends_with_ci(dest, ".obj");
        ME_CHECK(ply || obj, ME_ERR_BAD_ARG, "mesh destination must end in .obj or .ply: %s",
                 destination_path);
        const bool with_color = vertex_mode == ME_VERTEX_COLOR && vertex_colors;
        OutputScope out_scope(ctx, depth);
        const auto t_entry = std::chrono::steady_clock::now();
        if (ctx->write_behind && obj) {
            // Before any GPU work of this call: the pinned buffer it will use must be free, and a failed earlier write
            // is reported NOW (this call has produced nothing yet; the caller can repeat it).  A pending write to the
            // same destination is waited for as well -- a second fopen("wb") would truncate the file under it.
            join_pending_write(ctx, ctx->write_next);
            for (int k = 0; k < (int)ctx->write_slots.size(); ++k)
                if (ctx->write_slots[(size_t)k].active && ctx->write_slots[(size_t)k].dest == dest) join_pending_write(ctx, k);
        }

        // ---- GPU: IndexedMesh::new + remap_face + vertex coordinates
        const size_t nv = (size_t)width * height;
        const DeviceMesh mesh = build_mesh(ctx, depth, width, height, original_width, original_height);
        const int64_t nverts = mesh.nverts, nfaces = mesh.nfaces;
        int32_t* vindex = mesh.vindex;
        int32_t* faces_dev = mesh.faces;
        float *uv_dev = mesh.uv, *xyz_dev = mesh.xyz;
        // ---- OBJ: the text is formatted on the GPU too (obj_format.hip); one D2H copy into pinned memory, one file
        // write.  ME_OBJ_HOST_FORMAT=1 keeps the host formatter below (the same Ryu digits; A/B and PLY path).
        static const bool host_format = getenv("ME_OBJ_HOST_FORMAT") != nullptr;
        if (obj && !host_format) {
            static const bool timing = getenv("ME_OBJ_TIMING") != nullptr;  // diagnostic: the legs on stderr
            const auto now = [] { return std::chrono::steady_clock::now(); };
            const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            const auto t0 = now();
            const std::string stem = file_stem(dest);
            const DeviceText t = obj_text_on_device(ctx, mesh, width, height, stem, vertex_mode, vertex_colors);
            // write-behind: this call's text goes into the pinned buffer whose earlier write has finished (waited for
            // here, a failure of it reported here), and a host thread writes the file while the caller moves on
            const int slot = ctx->write_behind ? ctx->write_next : 0;
            if (ctx->write_behind) join_pending_write(ctx, slot);
            char* host = pinned_buf(ctx, (size_t)t.bytes, slot);
            ME_HIP(hipStreamSynchronize(ctx->stream));  // the legs are reported separately (me_last_mesh_timing)
            const auto t1 = now();
            ME_HIP(hipMemcpyAsync(host, t.dev, (size_t)t.bytes, hipMemcpyDeviceToHost, ctx->stream));
            ME_HIP(hipStreamSynchronize(ctx->stream));
            const auto t2 = now();
            if (ctx->write_behind) {
                me_ctx::WriteSlot& w = ctx->write_slots[(size_t)slot];
                const std::string src(source_path);
                const size_t nbytes = (size_t)t.bytes;
                const bool tex = vertex_mode == ME_VERTEX_TEXTURE;
                w.active = true, w.code = 0, w.msg.clear(), w.dest = dest;
                w.th = std::thread([&w, dest, stem, src, host, nbytes, tex]() {
                    try {
                        write_file_parallel(dest, host, nbytes);
                        if (tex) write_mtl(dest, stem, src.c_str());
                    } catch (const me::Error& e) {
                        w.code = e.code, w.msg = e.msg;
                    } catch (const std::exception& e) {
                        w.code = ME_ERR_IO, w.msg = e.what();
                    }
                });
                ctx->write_next = (ctx->write_next + 1) % ctx->write_behind;
            } else {
                write_file_parallel(dest, host, (size_t)t.bytes);
                if (vertex_mode == ME_VERTEX_TEXTURE) write_mtl(dest, stem, source_path);
            }
            const auto t3 = now();
            ctx->mesh_ms[0] = ms(t_entry, t0), ctx->mesh_ms[1] = ms(t0, t1), ctx->mesh_ms[2] = ms(t1, t2), ctx->mesh_ms[3] = ms(t2, t3);
            ctx->mesh_bytes = t.bytes;
            if (timing)
                fprintf(stderr, "me_output_mesh: %lld vertices, %lld faces, %lld bytes: mesh %.2f ms, format %.2f ms, D2H %.2f ms, "
                                "file %.2f ms\n", (long long)nverts, (long long)nfaces, (long long)t.bytes, ms(t_entry, t0), ms(t0, t1),
                        ms(t1, t2), ms(t2, t3));
            return ME_OK;
        }
        std::vector<float> uv((size_t)nverts * 2), xyz((size_t)nverts * 3);
        std::vector<int32_t> faces((size_t)nfaces * 3), vi;
        if (nverts) {
            ME_HIP(hipMemcpyAsync(uv.data(), uv_dev, uv.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            ME_HIP(hipMemcpyAsync(xyz.data(), xyz_dev, xyz.size() * 4, hipMemcpyDeviceToHost,
                                  ctx->stream));
        }
        if (nfaces)
            ME_HIP(hipMemcpyAsync(faces.data(), faces_dev, faces.size() * 4, hipMemcpyDeviceToHost,
                                  ctx->stream));
        std::vector<uint8_t> colors;  // per vertex id
        if (with_color) {
            vi.resize(nv);
            ME_HIP(hipMemcpyAsync(vi.data(), vindex, nv * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        ME_HIP(hipStreamSynchronize(ctx->stream));
        if (with_color) {
            std::vector<uint8_t> host_colors;
            const uint8_t* src = vertex_colors;
            if (is_device_ptr(vertex_colors)) {
                host_colors.resize(nv * 3);
                ME_HIP(hipMemcpy(host_colors.data(), vertex_colors, nv * 3, hipMemcpyDeviceToHost));
                src = host_colors.data();
            }
            colors.resize((size_t)nverts * 3);
            for (size_t i = 0; i < nv; ++i)
                if (vi[i] >= 0) memcpy(&colors[(size_t)vi[i] * 3], src + i * 3, 3);
        }

        FileSink w(dest);
        std::string& b = w.buf;
        if (obj) {
            const bool tex = vertex_mode == ME_VERTEX_TEXTURE;
            const std::string stem = file_stem(dest);
            if (tex) {  // output.rs:556-562
                b += "mtllib " + stem + ".mtl\n";
                b += "usemtl Textured\n";
            }
            if (tex)  // output.rs:592-602
                write_section(w, nverts, [&](int64_t i, std::string& b) {
                    b += "vt ";
                    put_f64(b, (double)uv[2 * i]);
                    b += ' ';
                    put_f64(b, 1.0 - (double)uv[2 * i + 1]);
                    b += '\n';
                });
            write_section(w, nverts, [&](int64_t i, std::string& b) {  // output.rs:566-590
                b += "v ";
                put_f64(b, (double)xyz[3 * i]);
                b += ' ';
                put_f64(b, (double)(-xyz[3 * i + 1]));
                b += ' ';
                put_f64(b, (double)(-xyz[3 * i + 2]));
                if (with_color)
                    for (int c = 0; c < 3; ++c) {
                        b += ' ';
                        put_f64(b, (double)colors[3 * i + c] / 255.0);
                    }
                b += '\n';
            });
            write_section(w, nfaces, [&](int64_t f, std::string& b) {  // output.rs:604-620
                b += 'f';
                for (int k = 0; k < 3; ++k) {
                    const unsigned long long idx = (unsigned long long)faces[3 * f + k] + 1;
                    b += ' ';
                    put_u64(b, idx);
                    if (tex) {
                        b += '/';
                        put_u64(b, idx);
                    }
                }
                b += '\n';
            });
            w.close();
            if (tex) write_mtl(dest, stem, source_path);
        } else {
            // output.rs:415-438
            b += "ply\nformat binary_big_endian 1.0\ncomment Matrix Eyes 3D surface\n";
            b += "element vertex " + std::to_string(nverts) + "\n";
            b += "property double x\nproperty double y\nproperty double z\n";
            if (vertex_mode == ME_VERTEX_COLOR)
                b += "property uchar red\nproperty uchar green\nproperty uchar blue\n";
            b += "element face " + std::to_string(nfaces) + "\n";
            b += "property list uchar int vertex_indices\nend_header\n";
            write_section(w, nverts, [&](int64_t i, std::string& b) {  // output.rs:440-458
                put_be64(b, (double)xyz[3 * i]);
                put_be64(b, (double)(-xyz[3 * i + 1]));
                put_be64(b, (double)(-xyz[3 * i + 2]));
                if (with_color) b.append((const char*)&colors[3 * i], 3);
            }, false);  // binary: a byte shuffle per item, not worth the second copy
            write_section(w, nfaces, [&](int64_t f, std::string& b) {  // output.rs:464-473
                b.push_back((char)3);
                for (int k = 0; k < 3; ++k) put_be32(b, (uint32_t)faces[3 * f + k]);
            }, false);
            w.close();
        }
    } catch (const me::Error& e) {
        ctx->last_error = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        ctx->last_error = std::string("internal: ") + e.what();
        return ME_ERR_IO;
    }
    return ME_OK;
}
