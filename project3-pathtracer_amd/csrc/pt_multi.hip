// pt_multi.hip -- several devices of one node behind one handle (single process, one host thread).
//
// The path shards by pixels: device k renders the k-th band of rows of the frame with its own
// context, stream and ray pools -- no communication while rendering, RNG keyed on the global pixel so the
// union of the bands is the single-device image bit for bit.  The only exchange is the gather of the bands:
// to the host (pt_multi_download_image) or into one device buffer over xGMI peer copies
// (pt_multi_gather_to_device).  All launches are asynchronous, so one host thread keeps every device busy.
// bench.py uses one process per GPU + RCCL instead (the driver's contract); this file gives the same
// sharding to a C++ host such as the reference's main.cpp through the shim (PT_DEVICES=0,1,2,...).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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
};

namespace {
void band(int height, int world, int rank, int *r0, int *r1)
{
    *r0 = (int)((long long)height * rank / world);
    *r1 = (int)((long long)height * (rank + 1) / world);
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
    int rc = pt_multi_synchronize(m);
    if (rc != PT_OK) return rc;
    char *frame = host_frame ? (char *)host_frame : (char *)device_frame;
    for (int k = 0; k < n; ++k) {
        void *tile = nullptr;
        rc = pt_image_device_pointer(m->ctx[(size_t)k], &tile);
        if (rc != PT_OK) return rc;
        if (hipSetDevice(m->device[(size_t)k]) != hipSuccess) return pt::fail(PT_ERR_HIP, "copy_strips: %s", hipGetErrorString(hipGetLastError()));
        const int nstrips = (H + S - 1) / S;
        int full = 0, tail_rows = 0, tail_strip = -1;
        for (int j = k; j < nstrips; j += n) {
            if ((j + 1) * S <= H) ++full;
            else { tail_rows = H - j * S; tail_strip = j; }
        }
        char *f0 = frame + (size_t)k * stripb;                      // first strip of device k in the frame
        const hipMemcpyKind kind = host_frame ? (to_frame ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice) : hipMemcpyDefault;
        hipError_t e = hipSuccess;
        if (full > 0) {
            if (to_frame) e = hipMemcpy2D(f0, stripb * (size_t)n, tile, stripb, stripb, (size_t)full, kind);
            else e = hipMemcpy2D(tile, stripb, f0, stripb * (size_t)n, stripb, (size_t)full, kind);
        }
        if (e == hipSuccess && tail_rows > 0) {
            char *ft = frame + (size_t)tail_strip * stripb, *tt = (char *)tile + (size_t)full * stripb;
            e = to_frame ? hipMemcpy(ft, tt, rowb * (size_t)tail_rows, kind) : hipMemcpy(tt, ft, rowb * (size_t)tail_rows, kind);
        }
        if (e != hipSuccess) return pt::fail(PT_ERR_HIP, "copy_strips: %s", hipGetErrorString(e));
        (void)dst_device;
    }
    return PT_OK;
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
    }
    *out = m;
    return PT_OK;
}

void pt_multi_destroy(pt_multi *m)
{
    if (!m) return;
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
    return PT_OK;
}

// bands -> one host frame (W*H*3 fp32, row-major): each context copies its band straight into its slice
int pt_multi_download_image(pt_multi *m, float *host_rgb)
{
    if (!m || !host_rgb || !m->have_cam) return pt::fail(PT_ERR_INVALID, "pt_multi_download_image: invalid argument or state");
    const int n = (int)m->ctx.size();
    if (m->strip > 0 && n > 1) return copy_strips(m, host_rgb, nullptr, 0, /*to_frame=*/true);
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
    if (m->strip > 0 && n > 1) return copy_strips(m, const_cast<float *>(host_rgb), nullptr, 0, /*to_frame=*/false);
    for (int k = 0; k < n; ++k) {
        int r0, r1;
        band(m->height, n, k, &r0, &r1);
        int rc = pt_upload_image(m->ctx[(size_t)k], host_rgb + (size_t)r0 * (size_t)m->width * 3);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

// bands -> one DEVICE frame on device `dst_device` (W*H*3 fp32): peer copies over xGMI, one per band, each on its
// source device's path to the destination -- a set of concurrent point-to-point transfers, not a ring.
int pt_multi_gather_to_device(pt_multi *m, void *device_rgb, int dst_device)
{
    if (!m || !device_rgb || !m->have_cam) return pt::fail(PT_ERR_INVALID, "pt_multi_gather_to_device: invalid argument or state");
    const int n = (int)m->ctx.size();
    int rc = pt_multi_synchronize(m);
    if (rc != PT_OK) return rc;
    if (m->strip > 0 && n > 1) return copy_strips(m, nullptr, device_rgb, dst_device, /*to_frame=*/true);
    std::vector<hipStream_t> streams((size_t)n, nullptr);
    // every exit below goes through `finish`: streams created so far are drained and destroyed on error paths too
    auto finish = [&](int code) {
        for (int k = 0; k < n; ++k) {
            if (!streams[(size_t)k]) continue;
            (void)hipSetDevice(m->device[(size_t)k]);
            const hipError_t e = hipStreamSynchronize(streams[(size_t)k]);
            (void)hipStreamDestroy(streams[(size_t)k]);
            streams[(size_t)k] = nullptr;
            if (e != hipSuccess && code == PT_OK) code = pt::fail(PT_ERR_HIP, "pt_multi_gather_to_device: %s", hipGetErrorString(e));
        }
        return code;
    };
    for (int k = 0; k < n; ++k) {
        int r0, r1;
        band(m->height, n, k, &r0, &r1);
        const size_t bytes = (size_t)(r1 - r0) * (size_t)m->width * 3 * sizeof(float);
        void *src = nullptr;
        rc = pt_image_device_pointer(m->ctx[(size_t)k], &src);
        if (rc != PT_OK) return finish(rc);
        char *dst = (char *)device_rgb + (size_t)r0 * (size_t)m->width * 3 * sizeof(float);
        if (hipSetDevice(m->device[(size_t)k]) != hipSuccess) return finish(pt::fail(PT_ERR_HIP, "pt_multi_gather_to_device: %s", hipGetErrorString(hipGetLastError())));
        if (hipStreamCreateWithFlags(&streams[(size_t)k], hipStreamNonBlocking) != hipSuccess) {
            streams[(size_t)k] = nullptr;
            return finish(pt::fail(PT_ERR_HIP, "pt_multi_gather_to_device: %s", hipGetErrorString(hipGetLastError())));
        }
        hipError_t e = (m->device[(size_t)k] == dst_device)
                           ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, streams[(size_t)k])
                           : hipMemcpyPeerAsync(dst, dst_device, src, m->device[(size_t)k], bytes, streams[(size_t)k]);
        if (e != hipSuccess) return finish(pt::fail(PT_ERR_HIP, "pt_multi_gather_to_device: %s", hipGetErrorString(e)));
    }
    return finish(PT_OK);
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
