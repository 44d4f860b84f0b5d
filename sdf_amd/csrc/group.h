// group.h — one sdf::SDF on several GPUs of one process: a group of handles, one per
// listed device.  Query points are independent (the reference self-schedules single indices over its
// threads, util.cpp:10-15), so a query is cut into contiguous ranges [r * ceil(N/G), ...), one per member,
// each served by its own host thread, staging buffers and stream over the member's own PCIe link (host
// pointers), or the caller hands every member its device-resident shard (`_device`).  No data-path
// collective.  What must agree on all members is the mesh, its structures and the ray-cast rotation R
// (sdf.cpp:43-48 draws it at random): member 0 — the leader — draws R and builds; the others follow.
//
// One build per update (the reference's update() is one build per object, sdf.cpp:31-73).  The leader
// uploads the vertices and builds the nanoflann-exact kd-tree on the host with ALL build threads.  A
// follower never reads the host arrays and never builds a tree: it copies the vertices and then the built
// tree arrays (kd nodes, split positions, permuted vertices, position -> vertex, group refs) from the
// leader's HBM into its own with hipMemcpyPeerAsync — between two GPUs of a node that is one hop over
// their direct xGMI link, 7 followers pull over 7 different links at once — ordered behind the leader's
// uploads by two events (UpdateSync), and runs the device-side part of the update (per-face precompute,
// column grids, leaf boxes, oriented boxes: deterministic kernels, the same bytes on every GPU) itself,
// the face part concurrently with the leader's host build.
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

// device-resident shards: member k answers counts[k] points at d_pts[k] into d_out[k] on streams[k]
int group_query_device(sdfhip_group* g, int kind, const float* const* d_pts, const int64_t* counts, int trunc,
                       void* const* d_out, void* const* streams) {
    if (!g) return fail(SDFHIP_E_INVALID, "null group");
    if (!d_pts || !counts || !d_out) return fail(SDFHIP_E_INVALID, "null shard arrays");
    std::lock_guard<std::mutex> lk(g->mu);
    for (size_t k = 0; k < g->members.size(); ++k) {
        if (counts[k] < 0) return fail(SDFHIP_E_INVALID, "negative point count for member %zu", k);
        if (counts[k] > 0 && (!d_pts[k] || !d_out[k]))
            return fail(SDFHIP_E_INVALID, "null points/output pointer for member %zu", k);
    }
    return group_parallel(g, [&](size_t k) -> int {
        if (counts[k] == 0) return 0;
        return run_query_dev(g->members[k], kind, d_pts[k], counts[k], trunc, d_out[k], streams ? streams[k] : nullptr);
    });
}

// A follower's update: everything the leader `L` built arrives device to device.  With `sync` the leader is
// updating right now on another thread (stages: UpdateSync); without, the leader is complete and idle.
int follow_update(sdfhip_ctx* h, sdfhip_ctx* L, UpdateSync* sync) {
    auto t0 = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard dg(h->device);
    if (!dg.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    hipStream_t s = h->stream;
    if (int rc = wait_for_queries(h, s)) return rc;
    auto leader_failed = [&](int rc) { return fail(rc, "the group's leader failed: %s", sync->err.c_str()); };
    auto pull = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyPeerAsync(dst, h->device, src, L->device, bytes, s) : hipSuccess;
    };
    // ---- stage 1: vertices, mesh AABB, rotation --------------------------------------------------------
    if (sync)
        if (int rc = sync->wait(1)) return leader_failed(rc);
    std::memcpy(h->aabb, L->aabb, sizeof(h->aabb));
    h->opt_fast_nn = L->opt_fast_nn;
    if (h->robust) std::memcpy(h->R, L->R, sizeof(h->R));  // (the leader's, as drawn or pinned for this update)
    if (sync) HIPC(hipStreamWaitEvent(s, sync->ev_verts, 0));
    HIPC(pull(h->d_verts, L->d_verts, sizeof(float) * 3 * (size_t)h->nv));
    h->last_kd_ms = 0.f;  // no host build here
    if (h->opt_fast_nn) {  // the GPU-built tree: nothing of it comes from the host, every member builds its own
        if (int rc = update_device_side(h)) return rc;
        if (int rc = build_fast_tree(h)) return rc;
        h->amb_c = 1.f, h->amb_c2 = 1.f;
        HIPC(hipStreamSynchronize(s));
        h->fast_active = true;
        h->last_update_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    h->fast_active = false;
    // per-face precompute and column grids need the vertices and R only: they run while the leader builds the tree
    if (int rc = update_device_side(h)) return rc;
    // ---- stage 2: the built tree -----------------------------------------------------------------------
    if (sync)
        if (int rc = sync->wait(2)) return leader_failed(rc);
    h->approx_ok = h->approx_pre = L->approx_pre;  // (fill_apx below finishes the decision as the leader's does)
    h->amb_c = L->amb_c, h->amb_c2 = L->amb_c2;
    h->kd_depth = L->kd_depth, h->kd_root = L->kd_root, h->kd_npos = L->kd_npos, h->kd_nnodes = L->kd_nnodes;
    const uint32_t ngroups = L->n_groups;
    const size_t nalloc = (size_t)h->kd_npos + 16;
    if (int rc = reserve_groups(h, ngroups)) return rc;
    if (int rc = reserve_strict_tree(h, h->kd_nnodes, nalloc)) return rc;
    if (sync) HIPC(hipStreamWaitEvent(s, sync->ev_tree, 0));
    HIPC(pull(h->d_kd_vind, L->d_kd_vind, sizeof(int32_t) * (size_t)h->kd_npos));
    HIPC(pull(h->d_kd_pts4, L->d_kd_pts4, sizeof(float4) * nalloc));
    HIPC(pull(h->d_kd_pn, L->d_kd_pn, sizeof(sdfk::KdNodeH) * (size_t)h->kd_nnodes));
    HIPC(pull(h->d_kd_split, L->d_kd_split, sizeof(uint32_t) * (size_t)h->kd_nnodes));
    HIPC(pull(h->d_kd_gref, L->d_kd_gref, sizeof(uint32_t) * (size_t)ngroups));
    k_leaf_boxes<<<(h->kd_nnodes + 255) / 256 + (h->kd_nnodes == 0), 256, 0, s>>>(
        h->d_kd_pn, h->kd_nnodes, h->kd_root, h->aabb[0], h->aabb[1], h->aabb[2], h->aabb[3], h->aabb[4], h->aabb[5],
        h->d_kd_lbox);
    if (int rc = build_obbs(h, ngroups)) return rc;
    if (int rc = fill_apx(h, ngroups, nalloc)) return rc;  // (synchronizes: the leader may start its next update once
                                                           // every follower is through)
    h->last_update_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// the whole group: leader and followers on one host thread each
int group_update_all(sdfhip_group* g) {
    sdfhip_ctx* h0 = g->members[0];
    if (g->members.size() == 1) return sdfhip_update(h0);
    UpdateSync sync;
    {
        DevGuard dg(h0->device);
        if (!dg.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h0->device);
        HIPC(hipEventCreateWithFlags(&sync.ev_verts, hipEventDisableTiming));
        if (hipEventCreateWithFlags(&sync.ev_tree, hipEventDisableTiming) != hipSuccess) {
            (void)hipEventDestroy(sync.ev_verts);
            return fail(SDFHIP_E_HIP, "hipEventCreate failed");
        }
    }
    const int rc = group_parallel(g, [&](size_t k) -> int {
        if (k > 0) return follow_update(g->members[k], h0, &sync);
        int rc0;
        {
            std::lock_guard<std::mutex> lk(h0->mu);
            DevGuard dg(h0->device);
            rc0 = !dg.ok ? fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h0->device)
                         : guarded([&]() -> int { return do_update(h0, false /* R: drawn by the caller */, &sync); });
        }
        if (rc0) sync.fail_with(rc0, g_err);  // release the followers whatever happened
        else sync.post(2);
        return rc0;
    });
    {
        DevGuard dg(h0->device);
        (void)hipEventDestroy(sync.ev_verts);
        (void)hipEventDestroy(sync.ev_tree);
    }
    return rc;
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
        // the leader first: a complete handle (its own build, all build threads) that draws the rotation
        for (int k = 0; k < ndev; ++k) {
            sdfhip_ctx* h = nullptr;
            // copy = 1: the leader owns the host copy (sdfhip_verts_mutable of the group); the others keep a
            // pointer to it for the accessors only — their updates never read host vertices
            const float* vsrc = k == 0 || !copy ? verts : g->members[0]->owned_verts.data();
            const uint32_t* fsrc = k == 0 || !copy ? faces : g->members[0]->owned_faces.data();
            if (int rc1 = create_handle(vsrc, nv, fsrc, nf, robust, k == 0 ? copy : 0, devices[k], &h, k == 0)) return rc1;
            g->members.push_back(h);
            if (k > 0 && devices[k] != devices[0]) {  // direct peer copies where the link allows them
                DevGuard dg(devices[k]);
                int can = 0;
                if (dg.ok && hipDeviceCanAccessPeer(&can, devices[k], devices[0]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(devices[0], 0);
                (void)hipGetLastError();  // ("already enabled" is fine; without peer access the copy is staged)
            }
        }
        if (ndev > 1)  // the followers take the leader's structures device to device, concurrently
            return group_parallel(g, [&](size_t k) -> int { return k == 0 ? 0 : follow_update(g->members[k], g->members[0], nullptr); });
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
    // the leader draws the new rotation (unless the caller pinned one); the followers copy it at stage 1
    sdfhip_ctx* h0 = g->members[0];
    if (h0->robust && !g->user_pinned) {
        std::lock_guard<std::mutex> l0(h0->mu);
        draw_rotation(h0);
        h0->R_pinned = true;  // (as drawn: every member updates with exactly this matrix)
    }
    return group_update_all(g);
}

int sdfhip_group_update_from_device(sdfhip_group_t* g, const float* d_verts, void* stream) {
    if (!g || !d_verts) return fail(SDFHIP_E_INVALID, "null group/verts");
    sdfhip_ctx* h0 = g->members[0];
    if (!h0->own)
        return fail(SDFHIP_E_STATE, "update_from_device needs an owning group (copy=1): the host copy is refreshed");
    {   // the leader's host copy first (the kd-tree is built from it); the followers never read it
        std::lock_guard<std::mutex> lk(g->mu);
        std::lock_guard<std::mutex> l0(h0->mu);
        DevGuard dg(h0->device);
        if (!dg.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h0->device);
        HIPC(hipMemcpyAsync(h0->owned_verts.data(), d_verts, sizeof(float) * 3 * (size_t)h0->nv, hipMemcpyDeviceToHost,
                            (hipStream_t)stream));
        HIPC(hipStreamSynchronize((hipStream_t)stream));
    }
    return sdfhip_group_update(g);
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
int sdfhip_group_sdf_device(sdfhip_group_t* g, const float* const* p, const int64_t* n, int trunc, float* const* out,
                            void* const* streams) {
    return guarded([&]() -> int { return group_query_device(g, Q_SDF, p, n, trunc, (void* const*)out, streams); });
}
int sdfhip_group_contains_device(sdfhip_group_t* g, const float* const* p, const int64_t* n, uint8_t* const* out,
                                 void* const* streams) {
    return guarded([&]() -> int { return group_query_device(g, Q_CONTAINS, p, n, 0, (void* const*)out, streams); });
}
int sdfhip_group_nn_device(sdfhip_group_t* g, const float* const* p, const int64_t* n, int32_t* const* out,
                           void* const* streams) {
    return guarded([&]() -> int { return group_query_device(g, Q_NN, p, n, 0, (void* const*)out, streams); });
}

}  // extern "C"
