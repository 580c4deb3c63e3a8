// The request path below ModelInfer (reference: inference_engine/src/model.cpp:1158-1328): a request runs on one free lane, or -- batch sharding -- as
// contiguous row slices on the shard replicas side by side; concurrent requests can be coalesced into one device batch (dynamic batcher).
#include "bridge_internal.h"

namespace ie_bridge {


// outputs by index, in graph-output order (bridge:787-813); never writes past the caller's dims array
void write_out_dims(TensorData* outputs, int num_outputs, const std::vector<ie::IoDesc>& odesc, int64_t rows) {
    for (int i = 0; i < num_outputs && size_t(i) < odesc.size(); ++i) {
        std::vector<int64_t> dims = odesc[size_t(i)].dims;
        if (rows > 0 && !dims.empty()) dims[0] = rows;
        const int cap = outputs[i].shape.dims ? outputs[i].shape.num_dims : 0;
        const int nd = int(dims.size());
        if (outputs[i].shape.dims) {
            const int nw = nd < cap ? nd : cap;
            for (int j = 0; j < nw; ++j) outputs[i].shape.dims[j] = dims[size_t(j)];
            outputs[i].shape.num_dims = nw;
        }
    }
}

// Device-side accounting behind ModelGetMetadata.description: what ran (planner FLOPs / bytes) and how long the device took.
void ModelObj::Account(ie::DeviceModel& d, const ie::PlanInstance& pi) {
    const double ms = d.last_forward_ms();
    if (ms <= 0) return;
    std::lock_guard<std::mutex> g(acct_mu);
    acct_ms += ms;
    acct_flops += pi.plan.total_flops;
    acct_bytes += pi.plan.total_bytes;
    acct_forwards += 1;
    acct_images += pi.plan.inputs.empty() || pi.plan.inputs[0].dims.empty() ? 0 : pi.plan.inputs[0].dims[0];
}

// Run one device batch described by gather/scatter segments over `rows` rows (rows == 0: the shapes are used as they are and the
// batch cannot be cut).  One lane, or -- allow_shard, num_shards > 1 and at least one row per replica -- contiguous row slices on
// all shard lanes at once (slice k on worker thread k-1, slice 0 on the calling thread).  Returns a COPY of the output descriptors
// of the plan that ran, taken while the lane is still held: once a lane is released any other request may take it and `Prepare` a
// new shape there, which can evict (free) this plan instance from the lane's bounded cache.
std::vector<ie::IoDesc> ModelObj::RunOnLanes(const std::vector<std::vector<int64_t>>& shapes, int64_t rows, bool allow_shard, const Segs& segs,
                                             bool* sharded) {
    const int S = num_shards;
    *sharded = false;
    if (!(allow_shard && S > 1 && rows >= S)) {
        const int k = pool.AcquireAny();
        if (k < 0) throw std::runtime_error("Model not loaded");
        try {
            ie::DeviceModel& D = *lanes[size_t(k)];
            ie::PlanInstance& pi = D.Prepare(shapes, false);
            D.InferHostSegments(pi, segs.first, segs.second);
            Account(D, pi);
            std::vector<ie::IoDesc> outs = pi.plan.outputs;
            pool.Release(k, 1);
            return outs;
        } catch (...) {
            pool.Release(k, 1);
            throw;
        }
    }
    pool.AcquireRange(S);
    std::vector<std::string> errs(static_cast<size_t>(S));
    std::vector<ie::PlanInstance*> pis(static_cast<size_t>(S), nullptr);
    auto run_slice = [&](int k) {
        try {
            ie::DeviceModel& D = *lanes[size_t(k)];
            const int64_t r0 = rows * k / S, r1 = rows * (k + 1) / S, nr = r1 - r0;
            std::vector<std::vector<int64_t>> sh = shapes;
            for (auto& x : sh) x[0] = nr;
            ie::PlanInstance& pi = D.Prepare(sh, false);
            pis[size_t(k)] = &pi;
            Segs mine;
            mine.first.resize(segs.first.size());
            mine.second.resize(segs.second.size());
            for (size_t i = 0; i < segs.first.size() && i < pi.plan.inputs.size(); ++i) {
                if (segs.first[i].empty()) continue;
                const size_t row_bytes = size_t(pi.plan.inputs[i].view.numel() / nr) * (segs.first[i][0].u8 ? 1 : sizeof(float));
                const size_t a = size_t(r0) * row_bytes, b = size_t(r1) * row_bytes;
                for (const auto& sg : segs.first[i]) {
                    const size_t lo = std::max(sg.dev_off, a), hi = std::min(sg.dev_off + sg.need, b);
                    if (hi <= lo) continue;
                    const size_t delta = lo - sg.dev_off;
                    ie::DeviceModel::InSeg n2{sg.host ? static_cast<const char*>(sg.host) + delta : nullptr,
                                              sg.have > delta ? std::min(sg.have - delta, hi - lo) : 0, hi - lo, lo - a, sg.u8};
                    mine.first[i].push_back(n2);
                }
            }
            for (size_t j = 0; j < segs.second.size() && j < pi.plan.outputs.size(); ++j) {
                const size_t row_bytes = size_t(pi.plan.outputs[j].view.numel() / nr) * sizeof(float);
                const size_t a = size_t(r0) * row_bytes, b = size_t(r1) * row_bytes;
                for (const auto& sg : segs.second[j]) {
                    const size_t nb = std::min(sg.cap, sg.need);
                    const size_t lo = std::max(sg.dev_off, a), hi = std::min(sg.dev_off + nb, b);
                    if (hi <= lo) continue;
                    mine.second[j].push_back({static_cast<char*>(sg.host) + (lo - sg.dev_off), hi - lo, hi - lo, lo - a});
                }
            }
            D.InferHostSegments(pi, mine.first, mine.second);
            Account(D, pi);
        } catch (const std::exception& e) {
            errs[size_t(k)] = e.what();
        } catch (...) {
            errs[size_t(k)] = "unknown error";
        }
    };
    for (int k = 1; k < S; ++k) workers.Submit(size_t(k - 1), [&run_slice, k] { run_slice(k); });
    run_slice(0);
    workers.Wait();
    std::vector<ie::IoDesc> outs0;
    if (pis[0]) outs0 = pis[0]->plan.outputs;
    pool.Release(0, S);
    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
    // the slices wrote only the bytes they produced: zero-fill whatever a caller buffer has beyond its result
    for (const auto& outs : segs.second)
        for (const auto& sg : outs) {
            const size_t nb = std::min(sg.cap, sg.need);
            if (sg.cap > nb) std::memset(static_cast<char*>(sg.host) + nb, 0, sg.cap - nb);
        }
    *sharded = true;
    return outs0;
}

void ModelObj::Execute(std::vector<Pending*>& batch) {
    std::shared_lock<std::shared_mutex> g(life);
    auto fail_all = [&](const std::string& msg) { for (auto* r : batch) { r->ok = false; r->err = msg; } };
    if (!loaded.load() || lanes.empty()) { fail_all("Model not loaded"); return; }
    try {
        const bool coalesced = !(batch.size() == 1 && !(batchable && max_batch > 1 && batch[0]->rows > 0));
        const size_t nin = info.inputs.size(), nout = info.outputs.size();
        Segs segs;
        segs.first.resize(nin);
        segs.second.resize(nout);
        // elements per row of every graph input / output come from the request's own shapes (inputs) and the model (outputs are
        // sized by the plan: the segment's `need` is clipped by InferHostSegments against the planned tensor)
        auto row_elems = [](const std::vector<int64_t>& sh) { size_t n = 1; for (size_t k = 1; k < sh.size(); ++k) n *= size_t(sh[k]); return n; };
        if (!coalesced) {
            Pending& r = *batch[0];
            const int64_t rows = r.shapes.empty() || r.shapes[0].empty() ? 0 : r.shapes[0][0];
            bool same_rows = rows > 0 && batchable;
            for (auto& sh : r.shapes) if (sh.empty() || sh[0] != rows) same_rows = false;
            bool sharded = false;
            if (same_rows && num_shards > 1 && rows >= num_shards) {
                // per-row sizes need the output row size: take it from the primary's plan for one row per shard ... the plan for the
                // slice is only known inside the slice, so describe outputs by the model's declared dims instead
                for (size_t i = 0; i < nin; ++i) {
                    const size_t rb = row_elems(r.shapes[i]) * (r.in_u8[i] ? 1 : sizeof(float));
                    segs.first[i].push_back({r.in_ptr[i], r.in_ptr[i] ? r.in_bytes[i] : 0, size_t(rows) * rb, 0, r.in_u8[i] != 0});
                }
                for (int j = 0; j < r.num_outputs && size_t(j) < nout; ++j) {
                    const TensorData& o = r.outputs[j];
                    if (o.data_type != DATATYPE_FLOAT32 || !o.data || o.data_size == 0) continue;
                    size_t re = 1;
                    bool known = !info.outputs[size_t(j)].dims.empty();
                    for (size_t k = 1; k < info.outputs[size_t(j)].dims.size(); ++k) {
                        if (info.outputs[size_t(j)].dims[k] <= 0) known = false;
                        else re *= size_t(info.outputs[size_t(j)].dims[k]);
                    }
                    if (!known) { same_rows = false; break; }
                    segs.second[size_t(j)].push_back({o.data, o.data_size, size_t(rows) * re * sizeof(float), 0});
                }
            }
            if (same_rows && num_shards > 1 && rows >= num_shards) {
                const std::vector<ie::IoDesc> outs = RunOnLanes(r.shapes, rows, true, segs, &sharded);
                write_out_dims(r.outputs, r.num_outputs, outs, rows);
                if (sharded) shard_calls.fetch_add(1);
                r.ok = true;
                return;
            }
            // ---- one request on one lane ----
            const int k = pool.AcquireAny();
            if (k < 0) { fail_all("Model not loaded"); return; }
            try {
                ie::DeviceModel& D = *lanes[size_t(k)];
                ie::PlanInstance& pi = D.Prepare(r.shapes, false);
                std::vector<void*> out_ptr;
                std::vector<size_t> out_bytes;
                for (int i = 0; i < r.num_outputs; ++i) {
                    const bool copy = r.outputs[i].data_type == DATATYPE_FLOAT32 && r.outputs[i].data && r.outputs[i].data_size > 0;
                    out_ptr.push_back(copy ? r.outputs[i].data : nullptr);
                    out_bytes.push_back(copy ? r.outputs[i].data_size : 0);
                }
                D.InferHost(pi, r.in_ptr, r.in_bytes, out_ptr, out_bytes, r.in_u8);
                Account(D, pi);
                write_out_dims(r.outputs, r.num_outputs, pi.plan.outputs, 0);
                pool.Release(k, 1);
            } catch (...) {
                pool.Release(k, 1);
                throw;
            }
            r.ok = true;
            return;
        }
        // ---- coalesced batch: rows of all callers back to back, padded up to a power-of-two bucket so only a handful of
        //      plans / hipGraphs ever exist; with shard replicas the bucket is cut over them like a single large request ----
        int64_t total = 0;
        for (auto* r : batch) total += r->rows;
        int64_t bucket = 1;
        while (bucket < total) bucket <<= 1;
        if (bucket > max_batch && total <= max_batch) bucket = max_batch;
        std::vector<std::vector<int64_t>> shapes = batch[0]->shapes;
        for (auto& sh : shapes) sh[0] = bucket;
        std::vector<size_t> out_row_bytes(nout, 0);
        bool out_known = true;
        for (size_t j = 0; j < nout; ++j) {
            size_t re = 1;
            if (info.outputs[j].dims.empty()) out_known = false;
            for (size_t k = 1; k < info.outputs[j].dims.size(); ++k) {
                if (info.outputs[j].dims[k] <= 0) out_known = false;
                else re *= size_t(info.outputs[j].dims[k]);
            }
            out_row_bytes[j] = re * sizeof(float);
        }
        if (!out_known) {      // output row size only known from a plan: take it from a one-lane plan of the bucket
            const int k = pool.AcquireAny();
            if (k < 0) { fail_all("Model not loaded"); return; }
            try {
                ie::PlanInstance& pi = lanes[size_t(k)]->Prepare(shapes, false);
                for (size_t j = 0; j < nout && j < pi.plan.outputs.size(); ++j) out_row_bytes[j] = size_t(pi.plan.outputs[j].view.numel() / bucket) * sizeof(float);
                pool.Release(k, 1);
            } catch (...) { pool.Release(k, 1); throw; }
        }
        int64_t row0 = 0;
        for (auto* r : batch) {
            for (size_t k = 0; k < nin; ++k) {
                const size_t rb = row_elems(shapes[k]) * sizeof(float);
                segs.first[k].push_back({r->in_ptr[k], r->in_bytes[k], size_t(r->rows) * rb, size_t(row0) * rb});
            }
            for (int j = 0; j < r->num_outputs && size_t(j) < nout; ++j) {
                const TensorData& o = r->outputs[j];
                if (o.data_type != DATATYPE_FLOAT32 || !o.data || o.data_size == 0) continue;
                segs.second[size_t(j)].push_back({o.data, o.data_size, size_t(r->rows) * out_row_bytes[size_t(j)], size_t(row0) * out_row_bytes[size_t(j)]});
            }
            row0 += r->rows;
        }
        // rows of the bucket beyond `total` stay whatever the input buffer held: they are padding whose results nobody reads
        bool sharded = false;
        const std::vector<ie::IoDesc> outs = RunOnLanes(shapes, bucket, out_known, segs, &sharded);
        device_batches.fetch_add(1);
        coalesced_requests.fetch_add(int64_t(batch.size()));
        if (sharded) shard_calls.fetch_add(1);
        for (auto* r : batch) {
            write_out_dims(r->outputs, r->num_outputs, outs, r->rows);
            r->ok = true;
        }
    } catch (const std::exception& e) {
        fail_all(std::string("ONNX inference error: ") + e.what());
    }
}

void ModelObj::RunBatched(Pending& req) {
    std::unique_lock<std::mutex> lk(bmu);
    queue.push_back(&req);
    bcv.notify_all();                                  // a waiting leader re-checks whether its batch is full
    while (!req.done) {
        if (leader_active) { bcv.wait(lk); continue; }
        leader_active = true;                          // this caller drives the next device batch
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(batch_window_us);
        auto queued_rows = [&] { int64_t n = 0; for (auto* r : queue) n += r->rows; return n; };
        while (queued_rows() < max_batch && bcv.wait_until(lk, deadline) != std::cv_status::timeout) {}
        std::vector<Pending*> batch;
        int64_t rows = 0;
        for (auto it = queue.begin(); it != queue.end();) {
            Pending* r = *it;
            bool compatible = batch.empty();
            if (!compatible) {
                compatible = rows + r->rows <= max_batch;
                for (size_t k = 0; k < r->shapes.size() && compatible; ++k)
                    compatible = std::equal(r->shapes[k].begin() + 1, r->shapes[k].end(), batch[0]->shapes[k].begin() + 1,
                                            batch[0]->shapes[k].end());
            }
            if (compatible) { batch.push_back(r); rows += r->rows; it = queue.erase(it); }
            else ++it;
        }
        lk.unlock();
        Execute(batch);
        lk.lock();
        for (auto* r : batch) r->done = true;
        leader_active = false;
        bcv.notify_all();
    }
}


}  // namespace ie_bridge
