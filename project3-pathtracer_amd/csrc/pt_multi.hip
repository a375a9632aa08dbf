// pt_multi.hip -- several devices of one node behind one handle (single process, one host thread).
//
// The path shards by pixels: device k renders the k-th band of rows of the frame with its own
// context, stream and ray pools -- no communication while rendering, RNG keyed on the global pixel so the
// union of the bands is the single-device image bit for bit.  The only exchange is the gather of the tiles:
// to the host (pt_multi_download_image) or into one device buffer over xGMI peer copies
// (pt_multi_gather_to_device[_async]).  All launches are asynchronous, so one host thread keeps every device busy.
// The gather: every source device has a copy stream of its own, ordered behind its render stream by an event (no host
// wait), the copies of all devices are enqueued back to back -- bands as one transfer each, interleaved strips as one
// strided 2-D transfer (+ a short last strip) -- and there is ONE join.  Peer access is looked up and enabled per
// (source, destination) pair; a pair without it is reported (pt_multi_peer_access) and served by hipMemcpyPeerAsync,
// which the runtime may stage through the host.
// bench.py uses one process per GPU + RCCL instead (the driver's contract); this file gives the same
// sharding to a C++ host such as the reference's main.cpp through the shim (PT_DEVICES=0,1,2,...).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <map>
#include <vector>

#include "pt_internal.h"

struct pt_multi {
    std::vector<pt_ctx *> ctx;
    std::vector<int> device;
    pt_options opt;
    pt_camera_data cam;
    bool have_cam = false;
    int height = 0, width = 0;
    int strip = 0;             // > 0: interleaved strips of this many rows instead of one band per device
    std::vector<hipStream_t> gstream;      // per context: copy stream of the gather, on that context's device
    std::vector<hipEvent_t> gevent;        // per context: recorded on its render stream, the copy stream waits for it
    std::map<std::pair<int, int>, int> peer;   // (source device, destination device) -> 1 peer access enabled, 0 none
    double gather_enqueue_ms = 0.0, gather_total_ms = 0.0;   // host clock of the last gather: until every copy was enqueued / until the join
    std::chrono::steady_clock::time_point gather_t0;
    bool gather_pending = false;
};

namespace {
void band(int height, int world, int rank, int *r0, int *r1)
{
    *r0 = (int)((long long)height * rank / world);
    *r1 = (int)((long long)height * (rank + 1) / world);
}

// Can `src` reach `dst`'s memory directly?  Looked up once per pair; access is enabled in both directions when it exists.
int peer_access(pt_multi *m, int src, int dst)
{
    if (src == dst) return 1;
    auto it = m->peer.find({src, dst});
    if (it != m->peer.end()) return it->second;
    int can = 0, ok = 0;
    if (hipDeviceCanAccessPeer(&can, src, dst) == hipSuccess && can) {
        ok = 1;
        for (int dir = 0; dir < 2; ++dir) {
            const int a = dir ? dst : src, b = dir ? src : dst;
            if (hipSetDevice(a) != hipSuccess) { ok = 0; break; }
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled && dir == 0) ok = 0;     // (the way back is a courtesy)
            (void)hipGetLastError();
        }
    }
    m->peer[{src, dst}] = ok;
    return ok;
}

// the copy streams continue behind everything enqueued on the render streams so far (no host wait)
int order_gather_streams(pt_multi *m)
{
    for (size_t k = 0; k < m->ctx.size(); ++k) {
        int rc = pt_record_event(m->ctx[k], (void *)m->gevent[k]);
        if (rc != PT_OK) return rc;
        if (hipSetDevice(m->device[k]) != hipSuccess || hipStreamWaitEvent(m->gstream[k], m->gevent[k], 0) != hipSuccess)
            return pt::fail(PT_ERR_HIP, "pt_multi gather: %s", hipGetErrorString(hipGetLastError()));
    }
    return PT_OK;
}

// ONE join of the copy streams
int join_gather_streams(pt_multi *m)
{
    int code = PT_OK;
    for (size_t k = 0; k < m->ctx.size(); ++k) {
        hipError_t e = hipSetDevice(m->device[k]);
        if (e == hipSuccess) e = hipStreamSynchronize(m->gstream[k]);
        if (e != hipSuccess && code == PT_OK) code = pt::fail(PT_ERR_HIP, "pt_multi gather: %s", hipGetErrorString(e));
    }
    if (m->gather_pending) {
        m->gather_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m->gather_t0).count();
        m->gather_pending = false;
    }
    return code;
}

int apply_options(pt_multi *m)
{
    const int n = (int)m->ctx.size();
    for (int k = 0; k < n; ++k) {
        pt_options o = m->opt;
        o.strip_rows = o.strip_world = o.strip_rank = 0;
        if (m->have_cam && n > 1) {
            if (m->strip > 0) { o.strip_rows = m->strip; o.strip_world = n; o.strip_rank = k; o.row_begin = o.row_end = 0; }
            else band(m->height, n, k, &o.row_begin, &o.row_end);
        }
        int rc = pt_set_options(m->ctx[(size_t)k], &o);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}
// Interleaved strips <-> one frame: device k's tile holds its strips packed, so tile -> frame is a strided 2-D copy
// (row = one strip, source pitch = strip bytes, destination pitch = n strips) plus the short last strip if it has one.
// host_frame != nullptr: frame in host memory; else frame on device `dst_device`.  to_frame=false: frame -> tiles (host).
int copy_strips(pt_multi *m, float *host_frame, void *device_frame, int dst_device, bool to_frame)
{
    const int n = (int)m->ctx.size(), S = m->strip, H = m->height;
    const size_t rowb = (size_t)m->width * 3 * sizeof(float), stripb = rowb * (size_t)S;
    int rc = order_gather_streams(m);
    if (rc != PT_OK) return rc;
    char *frame = host_frame ? (char *)host_frame : (char *)device_frame;
    for (int k = 0; k < n; ++k) {
        void *tile = nullptr;
        rc = pt_image_device_pointer(m->ctx[(size_t)k], &tile);
        if (rc != PT_OK) return rc;
        const int src_dev = m->device[(size_t)k];
        const bool direct = host_frame != nullptr || peer_access(m, src_dev, dst_device) != 0;
        if (hipSetDevice(src_dev) != hipSuccess) return pt::fail(PT_ERR_HIP, "copy_strips: %s", hipGetErrorString(hipGetLastError()));
        hipStream_t gs = m->gstream[(size_t)k];
        const int nstrips = (H + S - 1) / S;
        int full = 0, tail_rows = 0, tail_strip = -1;
        for (int j = k; j < nstrips; j += n) {
            if ((j + 1) * S <= H) ++full;
            else { tail_rows = H - j * S; tail_strip = j; }
        }
        char *f0 = frame + (size_t)k * stripb;                      // first strip of device k in the frame
        const hipMemcpyKind kind = host_frame ? (to_frame ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice) : hipMemcpyDefault;
        hipError_t e = hipSuccess;
        if (full > 0 && direct) {
            // device k's strips: one strided 2-D transfer (row = one strip; the frame's pitch is n strips)
            if (to_frame) e = hipMemcpy2DAsync(f0, stripb * (size_t)n, tile, stripb, stripb, (size_t)full, kind, gs);
            else e = hipMemcpy2DAsync(tile, stripb, f0, stripb * (size_t)n, stripb, (size_t)full, kind, gs);
        } else {
            // no peer access between the two devices: strip by strip through hipMemcpyPeerAsync (the runtime's staging path)
            for (int q = 0; q < full && e == hipSuccess; ++q)
                e = hipMemcpyPeerAsync(f0 + (size_t)q * stripb * (size_t)n, dst_device, (char *)tile + (size_t)q * stripb, src_dev, stripb, gs);
        }
        if (e == hipSuccess && tail_rows > 0) {
            char *ft = frame + (size_t)tail_strip * stripb, *tt = (char *)tile + (size_t)full * stripb;
            const size_t tb = rowb * (size_t)tail_rows;
            if (!direct) e = hipMemcpyPeerAsync(ft, dst_device, tt, src_dev, tb, gs);
            else e = to_frame ? hipMemcpyAsync(ft, tt, tb, kind, gs) : hipMemcpyAsync(tt, ft, tb, kind, gs);
        }
        if (e != hipSuccess) { (void)join_gather_streams(m); return pt::fail(PT_ERR_HIP, "copy_strips: %s", hipGetErrorString(e)); }
    }
    return PT_OK;
}

// tiles -> one device frame, enqueued on the copy streams (no join)
int enqueue_gather_to_device(pt_multi *m, void *device_rgb, int dst_device)
{
    const int n = (int)m->ctx.size();
    if (m->gather_pending) { int rc = join_gather_streams(m); if (rc != PT_OK) return rc; }      // (one gather in flight per handle)
    m->gather_t0 = std::chrono::steady_clock::now();
    m->gather_pending = true;
    int rc = PT_OK;
    if (m->strip > 0 && n > 1) rc = copy_strips(m, nullptr, device_rgb, dst_device, /*to_frame=*/true);
    else {
        rc = order_gather_streams(m);
        for (int k = 0; k < n && rc == PT_OK; ++k) {
            int r0, r1;
            band(m->height, n, k, &r0, &r1);
            const size_t bytes = (size_t)(r1 - r0) * (size_t)m->width * 3 * sizeof(float);
            void *src = nullptr;
            rc = pt_image_device_pointer(m->ctx[(size_t)k], &src);
            if (rc != PT_OK) break;
            char *dst = (char *)device_rgb + (size_t)r0 * (size_t)m->width * 3 * sizeof(float);
            const int src_dev = m->device[(size_t)k];
            const bool direct = peer_access(m, src_dev, dst_device) != 0;
            hipError_t e = hipSetDevice(src_dev);
            if (e == hipSuccess)
                e = direct ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, m->gstream[(size_t)k])
                           : hipMemcpyPeerAsync(dst, dst_device, src, src_dev, bytes, m->gstream[(size_t)k]);
            if (e != hipSuccess) rc = pt::fail(PT_ERR_HIP, "pt_multi_gather_to_device: %s", hipGetErrorString(e));
        }
    }
    m->gather_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m->gather_t0).count();
    if (rc != PT_OK) (void)join_gather_streams(m);
    return rc;
}
}  // namespace

extern "C" {

int pt_multi_create(const int *devices, int n, pt_multi **out)
{
    if (!devices || n < 1 || n > 64 || !out) return pt::fail(PT_ERR_INVALID, "pt_multi_create: invalid argument or state");
    *out = nullptr;
    pt_multi *m = new pt_multi();
    pt_default_options(&m->opt);
    for (int k = 0; k < n; ++k) {
        pt_ctx *c = nullptr;
        int rc = pt_create(devices[k], &c);
        if (rc != PT_OK) {
            for (pt_ctx *x : m->ctx) pt_destroy(x);
            delete m;
            return rc;
        }
        m->ctx.push_back(c);
        m->device.push_back(devices[k]);
        hipStream_t gs = nullptr;
        hipEvent_t ge = nullptr;
        if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&gs, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ge, hipEventDisableTiming) != hipSuccess) {
            const int code = pt::fail(PT_ERR_HIP, "pt_multi_create: %s", hipGetErrorString(hipGetLastError()));
            if (gs) (void)hipStreamDestroy(gs);
            pt_multi_destroy(m);
            return code;
        }
        m->gstream.push_back(gs);
        m->gevent.push_back(ge);
    }
    // peer access between the handle's devices: looked up and enabled now, reported by pt_multi_peer_access (never fatal)
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) (void)peer_access(m, devices[a], devices[b]);
    *out = m;
    return PT_OK;
}

void pt_multi_destroy(pt_multi *m)
{
    if (!m) return;
    for (size_t k = 0; k < m->gstream.size(); ++k) {
        (void)hipSetDevice(m->device[k]);
        (void)hipStreamSynchronize(m->gstream[k]);
        (void)hipStreamDestroy(m->gstream[k]);
        (void)hipEventDestroy(m->gevent[k]);
    }
    for (pt_ctx *c : m->ctx) pt_destroy(c);
    delete m;
}

int pt_multi_count(const pt_multi *m) { return m ? (int)m->ctx.size() : 0; }

// strip_rows > 0: device k renders strips k, k+n, k+2n, ... of strip_rows rows each (balances the devices when path
// lengths vary down the frame); 0: one contiguous band per device (default)
int pt_multi_set_strips(pt_multi *m, int strip_rows)
{
    if (!m || strip_rows < 0) return pt::fail(PT_ERR_INVALID, "pt_multi_set_strips: invalid argument or state");
    if (strip_rows > 0 && m->have_cam && (long long)strip_rows * (long long)m->ctx.size() > (long long)m->height + strip_rows - 1)
        return pt::fail(PT_ERR_INVALID, "pt_multi_set_strips: invalid argument or state");                     // some device would own no strip
    m->strip = strip_rows;
    return apply_options(m);
}

int pt_multi_set_options(pt_multi *m, const pt_options *o)
{
    if (!m || !o) return pt::fail(PT_ERR_INVALID, "pt_multi_set_options: invalid argument or state");
    m->opt = *o;
    return apply_options(m);
}

int pt_multi_set_scene(pt_multi *m, const pt_static_geom *geoms, int nG, const pt_material *mats, int nM)
{
    if (!m) return pt::fail(PT_ERR_INVALID, "pt_multi_set_scene: invalid argument or state");
    for (pt_ctx *c : m->ctx) {
        int rc = pt_set_scene(c, geoms, nG, mats, nM);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

int pt_multi_set_meshes(pt_multi *m, const pt_mesh *meshes, int n)
{
    if (!m) return pt::fail(PT_ERR_INVALID, "pt_multi_set_meshes: invalid argument or state");
    for (pt_ctx *c : m->ctx) {
        int rc = pt_set_meshes(c, meshes, n);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

int pt_multi_set_motion(pt_multi *m, const pt_static_geom *geoms_next, const pt_camera_data *cam_next, int slices, int rotat_units)
{
    if (!m) return pt::fail(PT_ERR_INVALID, "pt_multi_set_motion: invalid argument or state");
    for (pt_ctx *c : m->ctx) {
        int rc = pt_set_motion(c, geoms_next, cam_next, slices, rotat_units);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

int pt_multi_set_camera(pt_multi *m, const pt_camera_data *cam)
{
    if (!m || !cam) return pt::fail(PT_ERR_INVALID, "pt_multi_set_camera: invalid argument or state");
    const int H = (int)cam->resolution.y, W = (int)cam->resolution.x;
    if (H < (int)m->ctx.size() || W < 1) return pt::fail(PT_ERR_INVALID, "pt_multi_set_camera: invalid argument or state");          // every device needs at least one row
    if (m->strip > 0 && (long long)m->strip * (long long)m->ctx.size() > (long long)H + m->strip - 1) return pt::fail(PT_ERR_INVALID, "pt_multi_set_camera: invalid argument or state");
    m->cam = *cam;
    m->have_cam = true;
    m->height = H;
    m->width = W;
    for (pt_ctx *c : m->ctx) {
        int rc = pt_set_camera(c, cam);
        if (rc != PT_OK) return rc;
    }
    return apply_options(m);       // bands depend on the frame height
}

int pt_multi_band(const pt_multi *m, int k, int *row_begin, int *row_end)
{
    if (!m || !m->have_cam || k < 0 || k >= (int)m->ctx.size() || m->strip > 0) return pt::fail(PT_ERR_INVALID, "pt_multi_band: invalid argument or state");
    band(m->height, (int)m->ctx.size(), k, row_begin, row_end);
    return PT_OK;
}

int pt_multi_clear_image(pt_multi *m)
{
    if (!m) return pt::fail(PT_ERR_INVALID, "pt_multi_clear_image: invalid argument or state");
    for (pt_ctx *c : m->ctx) { int rc = pt_clear_image(c); if (rc != PT_OK) return rc; }
    return PT_OK;
}

int pt_multi_render(pt_multi *m, int iter_first, int iter_count)
{
    if (!m) return pt::fail(PT_ERR_INVALID, "pt_multi_render: invalid argument or state");
    for (pt_ctx *c : m->ctx) {                 // asynchronous on every device's own stream
        int rc = pt_render(c, iter_first, iter_count);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

int pt_multi_synchronize(pt_multi *m)
{
    if (!m) return pt::fail(PT_ERR_INVALID, "pt_multi_synchronize: invalid argument or state");
    for (pt_ctx *c : m->ctx) { int rc = pt_synchronize(c); if (rc != PT_OK) return rc; }
    return join_gather_streams(m);         // (a gather enqueued by pt_multi_gather_to_device_async)
}

// bands -> one host frame (W*H*3 fp32, row-major): each context copies its band straight into its slice
int pt_multi_download_image(pt_multi *m, float *host_rgb)
{
    if (!m || !host_rgb || !m->have_cam) return pt::fail(PT_ERR_INVALID, "pt_multi_download_image: invalid argument or state");
    const int n = (int)m->ctx.size();
    if (m->strip > 0 && n > 1) {
        int rc = copy_strips(m, host_rgb, nullptr, 0, /*to_frame=*/true);
        const int rj = join_gather_streams(m);
        return rc != PT_OK ? rc : rj;
    }
    for (int k = 0; k < n; ++k) {
        int r0, r1;
        band(m->height, n, k, &r0, &r1);
        int rc = pt_download_image(m->ctx[(size_t)k], host_rgb + (size_t)r0 * (size_t)m->width * 3);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

// host frame -> bands (resuming an accumulation, ref: src/raytraceKernel.cu:120)
int pt_multi_upload_image(pt_multi *m, const float *host_rgb)
{
    if (!m || !host_rgb || !m->have_cam) return pt::fail(PT_ERR_INVALID, "pt_multi_upload_image: invalid argument or state");
    const int n = (int)m->ctx.size();
    if (m->strip > 0 && n > 1) {
        int rc = pt_multi_synchronize(m);          // (nothing may still be rendering into the tiles)
        if (rc == PT_OK) rc = copy_strips(m, const_cast<float *>(host_rgb), nullptr, 0, /*to_frame=*/false);
        const int rj = join_gather_streams(m);
        return rc != PT_OK ? rc : rj;
    }
    for (int k = 0; k < n; ++k) {
        int r0, r1;
        band(m->height, n, k, &r0, &r1);
        int rc = pt_upload_image(m->ctx[(size_t)k], host_rgb + (size_t)r0 * (size_t)m->width * 3);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

// tiles -> one DEVICE frame on device `dst_device` (W*H*3 fp32): peer copies over xGMI, one (bands) or two (strips: a strided 2-D
// transfer + the short last strip) per source device, each on that device's copy stream behind its render stream -- a set
// of concurrent point-to-point transfers, not a ring.  _async returns when everything is enqueued (pt_multi_synchronize or the
// next gather joins); the plain form joins before it returns.
int pt_multi_gather_to_device_async(pt_multi *m, void *device_rgb, int dst_device)
{
    if (!m || !device_rgb || !m->have_cam) return pt::fail(PT_ERR_INVALID, "pt_multi_gather_to_device: invalid argument or state");
    return enqueue_gather_to_device(m, device_rgb, dst_device);
}

int pt_multi_gather_to_device(pt_multi *m, void *device_rgb, int dst_device)
{
    int rc = pt_multi_gather_to_device_async(m, device_rgb, dst_device);
    if (rc != PT_OK) return rc;
    return join_gather_streams(m);
}

// host clock of the last device gather: ms until every copy was enqueued, ms until the join (0 while it is still in flight)
int pt_multi_gather_times(pt_multi *m, double *enqueue_ms, double *total_ms)
{
    if (!m || !enqueue_ms || !total_ms) return pt::fail(PT_ERR_INVALID, "pt_multi_gather_times: invalid argument or state");
    *enqueue_ms = m->gather_enqueue_ms;
    *total_ms = m->gather_pending ? 0.0 : m->gather_total_ms;
    return PT_OK;
}

// 1: device `src_device` reaches `dst_device`'s memory directly (peer access enabled: the gather's copies cross xGMI),
// 0: it does not (hipMemcpyPeerAsync then stages the copy); the pair is looked up when first asked for
int pt_multi_peer_access(pt_multi *m, int src_device, int dst_device, int *direct)
{
    if (!m || !direct) return pt::fail(PT_ERR_INVALID, "pt_multi_peer_access: invalid argument or state");
    *direct = peer_access(m, src_device, dst_device);
    return PT_OK;
}

// sendImageToPBO for a single-device handle (the PBO is a device pointer of the GL device)
int pt_multi_send_image_to_pbo(pt_multi *m, pt_uchar4 *device_pbo)
{
    if (!m || m->ctx.size() != 1) return pt::fail(PT_ERR_INVALID, "pt_multi_send_image_to_pbo: invalid argument or state");
    return pt_send_image_to_pbo(m->ctx[0], device_pbo);
}

int pt_multi_get_stats(pt_multi *m, pt_stats *out)
{
    if (!m || !out) return pt::fail(PT_ERR_INVALID, "pt_multi_get_stats: invalid argument or state");
    memset(out, 0, sizeof *out);
    for (pt_ctx *c : m->ctx) {
        pt_stats s;
        int rc = pt_get_stats(c, &s);
        if (rc != PT_OK) return rc;
        if (s.iterations > out->iterations) out->iterations = s.iterations;
        out->ray_bounces += s.ray_bounces;
        for (int b = 0; b < PT_MAX_DEPTH; ++b) out->live_in[b] += s.live_in[b];
        if (s.gpu_ms > out->gpu_ms) out->gpu_ms = s.gpu_ms;        // devices run concurrently
        out->bounce_launches += s.bounce_launches;
        out->shadow_rays += s.shadow_rays;
    }
    return PT_OK;
}

}  // extern "C"
