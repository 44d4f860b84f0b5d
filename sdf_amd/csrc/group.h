// group.h — one sdf::SDF on several GPUs of one process: a group of handles, one per
// listed device, fed from the same host arrays.  Query points are independent (the reference
// self-schedules single indices over its threads, util.cpp:10-15), so a host-pointer query is cut into
// contiguous ranges [r * ceil(N/G), ...), one per member, each served by its own host thread, staging
// buffers and stream over the member's own PCIe link; results land in the caller's array range by
// range.  No data-path collective.  What must agree on all members is the mesh and the ray-cast
// rotation R (sdf.cpp:43-48 draws it at random): member 0 draws it, the others are pinned to it.
// Part of sdfhip.hip (one translation unit).
#pragma once

struct sdfhip_group {
    std::vector<sdfhip_ctx*> members;
    bool user_pinned = false;  // R was set by the caller: updates keep it
    std::mutex mu;
};

namespace {

// runs fn(member index) on one host thread per member (the caller's thread takes member 0)
template <class F>
int group_parallel(sdfhip_group* g, F&& fn) {
    const size_t G = g->members.size();
    std::vector<int> rc(G, 0);
    std::vector<std::string> err(G);
    std::vector<std::thread> th;
    auto one = [&](size_t k) {
        rc[k] = guarded([&]() -> int { return fn(k); });
        if (rc[k]) err[k] = g_err;  // g_err is thread local
    };
    for (size_t k = 1; k < G; ++k) th.emplace_back(one, k);
    one(0);
    for (auto& t : th) t.join();
    for (size_t k = 0; k < G; ++k)
        if (rc[k]) return fail(rc[k], "group member %zu (device %d): %s", k, g->members[k]->device, err[k].c_str());
    return 0;
}

int group_query(sdfhip_group* g, int kind, const float* pts, int64_t n, int trunc, void* out) {
    if (!g) return fail(SDFHIP_E_INVALID, "null group");
    if (n < 0) return fail(SDFHIP_E_INVALID, "negative point count");
    if (n == 0) return 0;
    if (!pts || !out) return fail(SDFHIP_E_INVALID, "null points/output pointer");
    std::lock_guard<std::mutex> lk(g->mu);
    const int64_t G = (int64_t)g->members.size();
    const int64_t per = (n + G - 1) / G;
    const size_t osz = kind == Q_CONTAINS ? 1 : 4;
    return group_parallel(g, [&](size_t k) -> int {
        const int64_t lo = std::min<int64_t>(n, (int64_t)k * per), hi = std::min<int64_t>(n, lo + per);
        if (hi <= lo) return 0;
        return run_query_host(g->members[k], kind, pts + 3 * lo, hi - lo, trunc, (char*)out + osz * (size_t)lo);
    });
}

}  // namespace

extern "C" {

int sdfhip_group_create(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, int robust, int copy,
                        const int* devices, int ndev, sdfhip_group_t** out) {
    if (!out) return fail(SDFHIP_E_INVALID, "out is null");
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return fail(SDFHIP_E_INVALID, "devices must list 1..64 device ordinals");
    sdfhip_group* g = nullptr;
    const int rc = guarded([&]() -> int {
        g = new sdfhip_group();
        // member 0 first: it draws the rotation the others are pinned to
        for (int k = 0; k < ndev; ++k) {
            sdfhip_ctx* h = nullptr;
            // copy = 1: member 0 owns the host copy (sdfhip_verts_mutable of the group), the others read it
            const float* vsrc = k == 0 || !copy ? verts : g->members[0]->owned_verts.data();
            const uint32_t* fsrc = k == 0 || !copy ? faces : g->members[0]->owned_faces.data();
            if (int rc1 = sdfhip_create(vsrc, nv, fsrc, nf, robust, k == 0 ? copy : 0, devices[k], &h)) return rc1;
            g->members.push_back(h);
            // the host threads of the kd-tree builds of one update share the cores
            h->opt_build_threads = std::max<int64_t>(1, h->opt_build_threads / ndev);
            if (k > 0 && robust)
                if (int rc1 = sdfhip_set_raycast_axes(h, g->members[0]->R, 1)) return rc1;
        }
        return 0;
    });
    if (rc) {
        const std::string keep = g_err;
        sdfhip_group_destroy(g);
        g_err = keep;
        return rc;
    }
    *out = g;
    return 0;
}

void sdfhip_group_destroy(sdfhip_group_t* g) {
    if (!g) return;
    for (sdfhip_ctx* h : g->members) sdfhip_destroy(h);
    delete g;
}

int sdfhip_group_size(const sdfhip_group_t* g) { return g ? (int)g->members.size() : 0; }
sdfhip_t* sdfhip_group_member(sdfhip_group_t* g, int k) {
    if (!g || k < 0 || k >= (int)g->members.size()) return nullptr;
    return g->members[(size_t)k];
}

int sdfhip_group_update(sdfhip_group_t* g) {
    if (!g) return fail(SDFHIP_E_INVALID, "null group");
    std::lock_guard<std::mutex> lk(g->mu);
    // member 0 draws the new rotation (unless the caller pinned one); the others take it at their own
    // update (pin = 2: stored now, applied by the update that follows), all of them concurrently
    sdfhip_ctx* h0 = g->members[0];
    if (h0->robust && !g->user_pinned) {
        std::lock_guard<std::mutex> l0(h0->mu);
        draw_rotation(h0);
        h0->R_pinned = true;  // (as drawn: every member updates with exactly this matrix)
    }
    for (size_t k = 1; k < g->members.size(); ++k)
        if (h0->robust)
            if (int rc = sdfhip_set_raycast_axes(g->members[k], h0->R, 2)) return rc;
    return group_parallel(g, [&](size_t k) -> int { return sdfhip_update(g->members[k]); });
}

int sdfhip_group_set_raycast_axes(sdfhip_group_t* g, const float R[9], int pin) {
    if (!g) return fail(SDFHIP_E_INVALID, "null group");
    std::lock_guard<std::mutex> lk(g->mu);
    g->user_pinned = pin != 0;
    if (!pin) return 0;  // the next update draws a fresh one for the whole group
    for (sdfhip_ctx* h : g->members)
        if (int rc = sdfhip_set_raycast_axes(h, R, 1)) return rc;
    return 0;
}
int sdfhip_group_get_raycast_axes(const sdfhip_group_t* g, float R[9]) {
    if (!g) return fail(SDFHIP_E_INVALID, "null group");
    return sdfhip_get_raycast_axes(g->members[0], R);
}

int sdfhip_group_sdf(sdfhip_group_t* g, const float* p, int64_t n, int trunc, float* out) {
    return guarded([&]() -> int { return group_query(g, Q_SDF, p, n, trunc, out); });
}
int sdfhip_group_contains(sdfhip_group_t* g, const float* p, int64_t n, uint8_t* out) {
    return guarded([&]() -> int { return group_query(g, Q_CONTAINS, p, n, 0, out); });
}
int sdfhip_group_nn(sdfhip_group_t* g, const float* p, int64_t n, int32_t* out) {
    return guarded([&]() -> int { return group_query(g, Q_NN, p, n, 0, out); });
}

}  // extern "C"
