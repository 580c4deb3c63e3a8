#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

#include "env.h"
#include "igemm_tiles.h"

namespace ie {

// Dense-layer fusion applies where launches are latency-bound: up to this many output pixels (IE_FUSE_MAX_M overrides; A/B switch).
static int64_t FuseMaxPixels(const Env& env) {
    const char* e = env.get("IE_FUSE_MAX_M");
    const long long x = e ? std::atoll(e) : 0;
    const int64_t v = x > 0 ? int64_t(x) : int64_t(8192);
    return v;
}
namespace {

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error(msg); }

struct Val {
    std::string name;
    std::vector<int64_t> dims;          // resolved ONNX dims
    int64_t n = 0, c = 0, h = 1, w = 1;
    int producer = -1;                  // LNode index, -1 = graph input
    bool is_input = false, is_output = false, input_nchw = false;
    int parent = -1;                    // concat parent value
    int64_t parent_off = 0;
    int root = -1;
    int64_t abs_off = 0;
    int buf = -1;
    bool dedicated = false;             // its buffer is never recycled
};

enum LKind { L_CONV, L_AFFINE, L_RELU, L_ADD, L_CONCAT, L_MAXPOOL, L_AVGPOOL, L_GAP, L_ALIAS, L_COPY };

struct LNode {
    LKind kind;
    std::string name;
    std::vector<int> in;
    int out = -1;
    int kh = 1, kw = 1, sh = 1, sw = 1, pt = 0, pl = 0, pb = 0, pr = 0;
    bool count_include_pad = false;
    std::vector<float> w, bias;         // conv: w packed [Cout][kh][kw][Cin]
    std::vector<float> s, t;            // affine
    bool has_pre = false, pre_relu = false, relu = false;
    std::vector<float> pre_s, pre_t;
    int res = -1;                       // conv: value added to the result before the ReLU (fused residual Add)
    bool dead = false;
};

struct Lowering {
    const OnnxModel& m;
    std::vector<Val> vals;
    std::map<std::string, int> val_of;
    std::vector<LNode> nodes;
    // initializers produced at plan time by folding shape-only ops (Unsqueeze / Squeeze / Reshape / Flatten / Identity) whose
    // input is itself a constant: model-zoo exports of Caffe BN+Scale pairs route the [C] scale and bias through Unsqueeze nodes
    std::map<std::string, OnnxTensor> derived;

    explicit Lowering(const OnnxModel& mm) : m(mm) {}

    int new_val(const std::string& name, const std::vector<int64_t>& dims) {
        if (val_of.count(name)) fail("ONNX graph error: value defined twice: " + name);
        Val v;
        v.name = name;
        v.dims = dims;
        if (dims.size() == 4) { v.n = dims[0]; v.c = dims[1]; v.h = dims[2]; v.w = dims[3]; }
        else if (dims.size() == 2) { v.n = dims[0]; v.c = dims[1]; }
        else if (dims.size() == 3) { v.n = dims[0]; v.c = dims[1]; v.h = dims[2]; }
        else if (dims.size() == 1) { v.n = 1; v.c = dims[0]; }
        else fail("unsupported tensor rank " + std::to_string(dims.size()) + " for value " + name);
        for (auto d : dims) if (d <= 0) fail("non-positive dimension in value " + name);
        vals.push_back(v);
        val_of[name] = int(vals.size()) - 1;
        return int(vals.size()) - 1;
    }
    int get_val(const std::string& name) const {
        auto it = val_of.find(name);
        if (it == val_of.end()) fail("ONNX graph error: undefined value: " + name);
        return it->second;
    }
    const OnnxTensor* init(const std::string& name) const {
        auto it = m.initializers.find(name);
        if (it != m.initializers.end()) return &it->second;
        auto jt = derived.find(name);
        return jt == derived.end() ? nullptr : &jt->second;
    }
    std::vector<int> consumers(int v) const {
        std::vector<int> out;
        for (size_t i = 0; i < nodes.size(); ++i)
            if (!nodes[i].dead)
                for (int x : nodes[i].in) if (x == v) { out.push_back(int(i)); break; }
        return out;
    }

    // A constant that broadcasts along the channel axis of `x` (numpy rules): scalar, [C], [1,C], [C,1,1], [1,C,1,1].
    // legacy_axis >= 0: opset < 7 broadcasting (attributes broadcast=1, axis=k): the constant's dims align with the activation's
    // dims starting at axis k instead of at the trailing end.
    bool per_channel_const(const OnnxTensor& t, const Val& x, std::vector<float>& out, int64_t legacy_axis = -1) const {
        if (t.dtype != ONNX_FLOAT && t.dtype != ONNX_DOUBLE && t.dtype != ONNX_FLOAT16) return false;
        int64_t n = t.numel();
        if (n == 1) { out.assign(size_t(x.c), t.f[0]); return true; }
        if (n != x.c) return false;
        size_t rank = x.dims.size();
        std::vector<int64_t> d = t.dims;
        if (legacy_axis >= 0) {
            if (size_t(legacy_axis) + d.size() > rank) return false;
            d.insert(d.begin(), size_t(legacy_axis), 1);
            while (d.size() < rank) d.push_back(1);
        }
        while (d.size() < rank) d.insert(d.begin(), 1);
        if (d.size() != rank) return false;
        for (size_t k = 0; k < rank; ++k) if (d[k] != (k == 1 ? x.c : 1)) {
            // rank-1 value [C] against rank-2 x [N,C] puts C on the last axis, which IS the channel axis
            return false;
        }
        out = t.f;
        return true;
    }
};

void conv_out_hw(int64_t h, int64_t w, const LNode& n, bool ceil_mode, int64_t& oh, int64_t& ow) {
    auto f = [&](int64_t x, int k, int s, int p0, int p1) {
        int64_t num = x + p0 + p1 - k;
        if (num < 0) fail("kernel larger than padded input in node " + n.name);
        return (ceil_mode ? (num + s - 1) / s : num / s) + 1;
    };
    oh = f(h, n.kh, n.sh, n.pt, n.pb);
    ow = f(w, n.kw, n.sw, n.pl, n.pr);
}

void read_window_attrs(const OnnxNode& on, LNode& n, int64_t h, int64_t w, bool is_conv, const OnnxTensor* wt) {
    std::vector<int64_t> ks = on.attr_ints("kernel_shape", {});
    if (ks.empty() && is_conv && wt) ks = {wt->dims[2], wt->dims[3]};
    if (ks.size() != 2) fail("node " + on.name + ": only 2-D kernels are supported");
    n.kh = int(ks[0]); n.kw = int(ks[1]);
    auto st = on.attr_ints("strides", {1, 1});
    n.sh = int(st[0]); n.sw = int(st[1]);
    auto dl = on.attr_ints("dilations", {1, 1});
    if (dl[0] != 1 || dl[1] != 1) fail("node " + on.name + ": dilations != 1 are not supported");
    auto pads = on.attr_ints("pads", {0, 0, 0, 0});
    if (pads.size() != 4) fail("node " + on.name + ": pads must have 4 entries");
    n.pt = int(pads[0]); n.pl = int(pads[1]); n.pb = int(pads[2]); n.pr = int(pads[3]);
    auto ap = on.attrs.find("auto_pad");
    if (ap != on.attrs.end() && !ap->second.s.empty() && ap->second.s != "NOTSET") {
        const std::string& mode = ap->second.s;
        if (mode == "VALID") n.pt = n.pl = n.pb = n.pr = 0;
        else if (mode == "SAME_UPPER" || mode == "SAME_LOWER") {
            auto same = [&](int64_t x, int k, int s, int& p0, int& p1) {
                int64_t o = (x + s - 1) / s;
                int64_t tot = std::max<int64_t>((o - 1) * s + k - x, 0);
                int64_t a = tot / 2, b = tot - a;
                if (mode == "SAME_UPPER") { p0 = int(a); p1 = int(b); } else { p0 = int(b); p1 = int(a); }
            };
            same(h, n.kh, n.sh, n.pt, n.pb);
            same(w, n.kw, n.sw, n.pl, n.pr);
        } else fail("node " + on.name + ": unsupported auto_pad " + mode);
    }
}

int choose_tile(int64_t M, int64_t N) {
    // Estimated time = rounds over the 256 CUs x tile area / tile efficiency.  Larger tiles reuse operands
    // better (higher MFMA duty); smaller ones fill the chip when M*N is small.  A lone workgroup per CU
    // cannot hide its own staging latency, so under-filled grids are charged a lower duty.
    static const double eff[kNumIgemmBaseTiles] = {1.00, 0.92, 0.78, 0.82, 0.62, 0.40, 0.84};
    int best = -1;
    double best_cost = 0;
    for (int t = 0; t < kNumIgemmBaseTiles; ++t) {
        const IgemmTile& T = kIgemmTiles[t];
        if (T.bn > 32 && N <= 32) continue;             // do not waste MFMA columns on zero padding
        if (T.bn > 64 && N <= 64) continue;
        double wgs = double((M + T.bm - 1) / T.bm) * double((N + T.bn - 1) / T.bn);
        double per_cu = wgs / 256.0;
        double rounds = std::max(1.0, 0.5 * (per_cu + std::ceil(per_cu)));
        double duty = wgs >= 512 ? 0.9 : (wgs >= 256 ? 0.75 : 0.6);
        double cost = rounds * double(T.bm) * T.bn / (eff[t] * duty);
        if (best < 0 || cost < best_cost) { best = t; best_cost = cost; }
    }
    return best;
}

}  // namespace

ModelInfo DescribeModel(const OnnxModel& m) {
    ModelInfo info;
    info.inputs = m.inputs;
    info.outputs = m.outputs;
    // EstimateModelMemoryUsage (model.cpp:979-1035): sum of I/O tensor bytes over positive dims + 10 MiB.
    auto elem = [](int t) -> size_t {
        switch (t) {
            case ONNX_FLOAT: case ONNX_INT32: return 4;
            case ONNX_INT64: return 8;
            case ONNX_UINT8: case ONNX_INT8: case ONNX_BOOL: return 1;
            case ONNX_FLOAT16: return 2;
            default: return 4;
        }
    };
    size_t total = 0;
    for (auto* list : {&m.inputs, &m.outputs})
        for (auto& vi : *list) {
            size_t n = 1;
            for (auto d : vi.dims) if (d > 0) n *= size_t(d);
            total += n * elem(vi.elem_type);
        }
    info.memory_usage_bytes = total + size_t(10) * 1024 * 1024;
    return info;
}

Plan BuildPlan(const OnnxModel& m, const std::vector<std::vector<int64_t>>& input_shapes, Precision precision, bool f8_fusions) {
    const Env env = Env::Read();          // the planner's switches, read once per plan build (load / prepare time)
    Lowering L(m);
    Plan plan;
    plan.precision = precision;
    if (input_shapes.size() != m.inputs.size())
        fail("Expected " + std::to_string(m.inputs.size()) + " inputs, got " + std::to_string(input_shapes.size()));

    // ---- graph inputs ---------------------------------------------------------------------------
    for (size_t i = 0; i < m.inputs.size(); ++i) {
        const auto& vi = m.inputs[i];
        const auto& got = input_shapes[i];
        if (vi.elem_type != ONNX_FLOAT) fail("Unsupported data type for input: " + vi.name);
        if (!vi.dims.empty()) {
            if (got.size() != vi.dims.size())
                fail("Invalid rank for input: " + vi.name + " Got: " + std::to_string(got.size()) +
                     " Expected: " + std::to_string(vi.dims.size()));
            for (size_t k = 0; k < got.size(); ++k)
                if (vi.dims[k] > 0 && vi.dims[k] != got[k])
                    fail("Got invalid dimensions for input: " + vi.name + " for the following indices index: " +
                         std::to_string(k) + " Got: " + std::to_string(got[k]) + " Expected: " + std::to_string(vi.dims[k]));
        }
        int v = L.new_val(vi.name, got);
        L.vals[v].is_input = true;
        L.vals[v].input_nchw = (L.vals[v].h * L.vals[v].w > 1);
    }

    // ---- ONNX nodes -> logical nodes with shape inference ---------------------------------------
    for (const auto& on : m.nodes) {
        if (on.outputs.empty() || on.inputs.empty()) fail("node " + on.name + " (" + on.op + ") has no inputs/outputs");
        LNode n;
        n.name = on.name.empty() ? on.outputs[0] : on.name;
        const std::string& op = on.op;
        auto in_val = [&](size_t k) { return L.get_val(on.inputs[k]); };
        auto act_input = [&](size_t k) { return k < on.inputs.size() && !on.inputs[k].empty() && !L.init(on.inputs[k]); };
        std::vector<int64_t> odims;

        // ---- shape-only ops on constants fold into a derived initializer (nothing is emitted) ----
        if ((op == "Unsqueeze" || op == "Squeeze" || op == "Reshape" || op == "Flatten" || op == "Identity") && L.init(on.inputs[0])) {
            OnnxTensor t = *L.init(on.inputs[0]);
            auto axes_of = [&]() {
                std::vector<int64_t> ax = on.attr_ints("axes", {});
                if (ax.empty() && on.inputs.size() > 1 && !on.inputs[1].empty()) {       // opset >= 13: axes is an input
                    const OnnxTensor* at = L.init(on.inputs[1]);
                    if (!at) fail(op + " " + n.name + ": axes must be an initializer");
                    ax = at->i;
                }
                return ax;
            };
            if (op == "Unsqueeze") {
                std::vector<int64_t> ax = axes_of();
                if (ax.empty()) fail("Unsqueeze " + n.name + ": axes are required");
                const int64_t orank = int64_t(t.dims.size() + ax.size());
                for (auto& a : ax) { if (a < 0) a += orank; if (a < 0 || a >= orank) fail("Unsqueeze " + n.name + ": axis out of range"); }
                std::sort(ax.begin(), ax.end());
                std::vector<int64_t> nd;
                size_t src = 0;
                for (int64_t k = 0; k < orank; ++k) {
                    if (std::binary_search(ax.begin(), ax.end(), k)) nd.push_back(1);
                    else nd.push_back(t.dims.at(src++));
                }
                t.dims = nd;
            } else if (op == "Squeeze") {
                std::vector<int64_t> ax = axes_of();
                const int64_t rank = int64_t(t.dims.size());
                for (auto& a : ax) if (a < 0) a += rank;
                std::vector<int64_t> nd;
                for (int64_t k = 0; k < rank; ++k) {
                    const bool listed = std::find(ax.begin(), ax.end(), k) != ax.end();
                    if (listed && t.dims[size_t(k)] != 1) fail("Squeeze " + n.name + ": cannot squeeze a dimension of size != 1");
                    if (ax.empty() ? t.dims[size_t(k)] != 1 : !listed) nd.push_back(t.dims[size_t(k)]);
                }
                t.dims = nd;
            } else if (op == "Reshape") {
                const OnnxTensor* shp = on.inputs.size() > 1 ? L.init(on.inputs[1]) : nullptr;
                std::vector<int64_t> d = shp ? shp->i : on.attr_ints("shape", {});
                if (d.empty() && t.numel() != 1) fail("Reshape " + n.name + ": shape must be an initializer");
                int64_t known = 1, neg = -1;
                for (size_t k = 0; k < d.size(); ++k) {
                    if (d[k] == 0 && k < t.dims.size()) d[k] = t.dims[k];
                    if (d[k] == -1) neg = int64_t(k); else known *= d[k];
                }
                if (neg >= 0 && known > 0) d[size_t(neg)] = t.numel() / known;
                int64_t tot = 1; for (auto v : d) tot *= v;
                if (tot != t.numel()) fail("Reshape " + n.name + ": element count mismatch");
                t.dims = d;
            } else if (op == "Flatten") {
                int64_t ax = on.attr_i("axis", 1);
                if (ax < 0) ax += int64_t(t.dims.size());
                int64_t a = 1, b = 1;
                for (size_t k = 0; k < t.dims.size(); ++k) (int64_t(k) < ax ? a : b) *= t.dims[k];
                t.dims = {a, b};
            }
            t.name = on.outputs[0];
            if (L.derived.count(t.name) || m.initializers.count(t.name) || L.val_of.count(t.name))
                fail("ONNX graph error: value defined twice: " + t.name);
            L.derived[t.name] = std::move(t);
            continue;
        }

        if (op == "Conv") {
            if (!act_input(0)) fail("Conv " + n.name + ": constant input is not supported");
            const OnnxTensor* w = L.init(on.inputs.at(1));
            if (!w || w->dims.size() != 4) fail("Conv " + n.name + ": weights must be a 4-D initializer");
            if (on.attr_i("group", 1) != 1) fail("Conv " + n.name + ": group != 1 is not supported");
            int x = in_val(0);
            const Val& X = L.vals[x];
            if (X.dims.size() != 4) fail("Conv " + n.name + ": input must be 4-D");
            int64_t co = w->dims[0], ci = w->dims[1];
            if (ci != X.c) fail("Conv " + n.name + ": input channels " + std::to_string(X.c) + " != weight channels " + std::to_string(ci));
            n.kind = L_CONV;
            read_window_attrs(on, n, X.h, X.w, true, w);
            if (n.kh != w->dims[2] || n.kw != w->dims[3]) fail("Conv " + n.name + ": kernel_shape does not match weights");
            n.w.resize(size_t(co * ci * n.kh * n.kw));
            for (int64_t o = 0; o < co; ++o)
                for (int64_t c = 0; c < ci; ++c)
                    for (int ky = 0; ky < n.kh; ++ky)
                        for (int kx = 0; kx < n.kw; ++kx)
                            n.w[size_t(((o * n.kh + ky) * n.kw + kx) * ci + c)] =
                                w->f[size_t(((o * ci + c) * n.kh + ky) * n.kw + kx)];
            if (on.inputs.size() > 2 && !on.inputs[2].empty()) {
                const OnnxTensor* b = L.init(on.inputs[2]);
                if (!b || b->numel() != co) fail("Conv " + n.name + ": bias must be a [Cout] initializer");
                n.bias = b->f;
            }
            int64_t oh, ow;
            conv_out_hw(X.h, X.w, n, false, oh, ow);
            n.in = {x};
            odims = {X.n, co, oh, ow};
        } else if (op == "MatMul" || op == "Gemm") {
            if (!act_input(0)) fail(op + " " + n.name + ": constant first operand is not supported");
            const OnnxTensor* b = L.init(on.inputs.at(1));
            if (!b || b->dims.size() != 2) fail(op + " " + n.name + ": second operand must be a 2-D initializer");
            int x = in_val(0);
            const Val& X = L.vals[x];
            if (X.h * X.w != 1) fail(op + " " + n.name + ": input must be [N, K]");
            bool transB = op == "Gemm" && on.attr_i("transB", 0) != 0;
            if (op == "Gemm" && on.attr_i("transA", 0) != 0) fail("Gemm " + n.name + ": transA is not supported");
            float alpha = op == "Gemm" ? on.attr_f("alpha", 1.f) : 1.f;
            float beta = op == "Gemm" ? on.attr_f("beta", 1.f) : 1.f;
            int64_t K = transB ? b->dims[1] : b->dims[0], N = transB ? b->dims[0] : b->dims[1];
            if (K != X.c) fail(op + " " + n.name + ": inner dimensions do not match");
            n.kind = L_CONV;
            n.w.resize(size_t(N * K));
            for (int64_t o = 0; o < N; ++o)
                for (int64_t k = 0; k < K; ++k)
                    n.w[size_t(o * K + k)] = alpha * (transB ? b->f[size_t(o * K + k)] : b->f[size_t(k * N + o)]);
            if (op == "Gemm" && on.inputs.size() > 2 && !on.inputs[2].empty()) {
                const OnnxTensor* c = L.init(on.inputs[2]);
                if (!c) fail("Gemm " + n.name + ": C must be an initializer");
                if (c->numel() == N) n.bias = c->f;
                else if (c->numel() == 1) n.bias.assign(size_t(N), c->f[0]);
                else fail("Gemm " + n.name + ": C must broadcast over rows");
                for (auto& v : n.bias) v *= beta;
            }
            n.in = {x};
            odims = {X.n, N};
            if (X.dims.size() == 1) odims = {N};
        } else if (op == "BatchNormalization") {
            int x = in_val(0);
            const Val& X = L.vals[x];
            const OnnxTensor *g = L.init(on.inputs.at(1)), *be = L.init(on.inputs.at(2)), *mu = L.init(on.inputs.at(3)),
                             *var = L.init(on.inputs.at(4));
            if (!g || !be || !mu || !var) fail("BatchNormalization " + n.name + ": parameters must be initializers");
            if (g->numel() != X.c || be->numel() != X.c || mu->numel() != X.c || var->numel() != X.c)
                fail("BatchNormalization " + n.name + ": parameter size != channels");
            double eps = on.attr_f("epsilon", 1e-5f);
            n.kind = L_AFFINE;
            n.s.resize(size_t(X.c)); n.t.resize(size_t(X.c));
            for (int64_t c = 0; c < X.c; ++c) {
                double s = double(g->f[c]) / std::sqrt(double(var->f[c]) + eps);
                n.s[c] = float(s);
                n.t[c] = float(double(be->f[c]) - double(mu->f[c]) * s);
            }
            n.in = {x};
            odims = X.dims;
        } else if (op == "Relu") {
            n.kind = L_RELU;
            n.in = {in_val(0)};
            odims = L.vals[n.in[0]].dims;
        } else if (op == "Add" || op == "Mul") {
            bool a0 = act_input(0), a1 = act_input(1);
            if (a0 && a1) {
                if (op == "Mul") fail("Mul of two activations is not supported (node " + n.name + ")");
                int a = in_val(0), b = in_val(1);
                if (L.vals[a].dims != L.vals[b].dims) fail("Add " + n.name + ": broadcasting between activations is not supported");
                n.kind = L_ADD;
                n.in = {a, b};
                odims = L.vals[a].dims;
            } else if (a0 || a1) {
                int x = in_val(a0 ? 0 : 1);
                const OnnxTensor* c = L.init(on.inputs[a0 ? 1 : 0]);
                std::vector<float> pc;
                // rank-1 constant against a rank-2 activation aligns with the last (= channel) axis
                const Val& X = L.vals[x];
                // opset < 7 (Caffe2-era exports): Add/Mul carry broadcast=1 and an axis that places the constant's dims inside the
                // activation's (axis=1 with a [C] constant = per channel); without an axis the constant aligns with the trailing dims
                int64_t legacy_axis = -1;
                if (m.opset > 0 && m.opset < 7 && on.attr_i("broadcast", 0) != 0 && on.attrs.count("axis")) {
                    legacy_axis = on.attr_i("axis", 0);
                    if (legacy_axis < 0) legacy_axis += int64_t(X.dims.size());
                }
                if (m.opset > 0 && m.opset < 7 && on.attr_i("broadcast", 0) == 0 && c->dims != X.dims && c->numel() != 1)
                    fail(op + " " + n.name + ": operand shapes differ and the opset-" + std::to_string(m.opset) + " broadcast attribute is not set");
                bool ok = L.per_channel_const(*c, X, pc, legacy_axis);
                if (!ok && X.dims.size() == 2 && c->dims.size() == 1 && c->numel() == X.c) { pc = c->f; ok = true; }
                if (!ok && X.dims.size() == 2 && c->dims.size() == 2 && c->dims[0] == 1 && c->dims[1] == X.c) { pc = c->f; ok = true; }
                if (!ok) fail(op + " " + n.name + ": constant operand must broadcast per channel");
                n.kind = L_AFFINE;
                if (op == "Add") { n.s.assign(size_t(X.c), 1.f); n.t = pc; }
                else { n.s = pc; n.t.assign(size_t(X.c), 0.f); }
                n.in = {x};
                odims = X.dims;
            } else fail(op + " " + n.name + ": constant folding of two initializers is not supported");
        } else if (op == "Concat") {
            int64_t axis = on.attr_i("axis", 1);
            n.kind = L_CONCAT;
            for (size_t k = 0; k < on.inputs.size(); ++k) {
                if (!act_input(k)) fail("Concat " + n.name + ": constant inputs are not supported");
                n.in.push_back(in_val(k));
            }
            const Val& X0 = L.vals[n.in[0]];
            if (axis < 0) axis += int64_t(X0.dims.size());
            if (axis != 1) fail("Concat " + n.name + ": only axis=1 (channels) is supported");
            odims = X0.dims;
            int64_t ctot = 0;
            for (int v : n.in) {
                const Val& X = L.vals[v];
                if (X.dims.size() != X0.dims.size()) fail("Concat " + n.name + ": rank mismatch");
                for (size_t k = 0; k < X.dims.size(); ++k)
                    if (k != 1 && X.dims[k] != X0.dims[k]) fail("Concat " + n.name + ": shape mismatch");
                ctot += X.c;
            }
            odims[1] = ctot;
        } else if (op == "MaxPool" || op == "AveragePool") {
            int x = in_val(0);
            const Val& X = L.vals[x];
            if (X.dims.size() != 4) fail(op + " " + n.name + ": input must be 4-D");
            if (on.outputs.size() > 1 && !on.outputs[1].empty()) fail("MaxPool " + n.name + ": Indices output is not supported");
            n.kind = op == "MaxPool" ? L_MAXPOOL : L_AVGPOOL;
            read_window_attrs(on, n, X.h, X.w, false, nullptr);
            n.count_include_pad = on.attr_i("count_include_pad", 0) != 0;
            int64_t oh, ow;
            conv_out_hw(X.h, X.w, n, on.attr_i("ceil_mode", 0) != 0, oh, ow);
            n.in = {x};
            odims = {X.n, X.c, oh, ow};
        } else if (op == "GlobalAveragePool") {
            int x = in_val(0);
            const Val& X = L.vals[x];
            if (X.dims.size() != 4) fail("GlobalAveragePool " + n.name + ": input must be 4-D");
            n.kind = L_GAP;
            n.in = {x};
            odims = {X.n, X.c, 1, 1};
        } else if (op == "Unsqueeze") {
            // [N,C] -> [N,C,1,1]: storage is unchanged when only trailing unit axes are added
            int x = in_val(0);
            const Val& X = L.vals[x];
            std::vector<int64_t> ax = on.attr_ints("axes", {});
            if (ax.empty() && on.inputs.size() > 1) { const OnnxTensor* at = L.init(on.inputs[1]); if (at) ax = at->i; }
            const int64_t orank = int64_t(X.dims.size() + ax.size());
            for (auto& a : ax) if (a < 0) a += orank;
            std::sort(ax.begin(), ax.end());
            if (ax.empty() || X.h * X.w != 1 || X.dims.size() < 2 || ax[0] < int64_t(X.dims.size()) || orank > 4)
                fail("Unsqueeze " + n.name + ": only trailing unit axes on [N,C] tensors are supported");
            n.kind = L_ALIAS;
            n.in = {x};
            odims = X.dims;
            while (int64_t(odims.size()) < orank) odims.push_back(1);
        } else if (op == "Flatten" || op == "Reshape" || op == "Identity" || op == "Dropout" || op == "Squeeze") {
            int x = in_val(0);
            const Val& X = L.vals[x];
            n.kind = L_ALIAS;
            n.in = {x};
            if (op == "Identity" || op == "Dropout") odims = X.dims;
            else {
                // Storage is NHWC: a reshape is a pure alias only when H*W == 1 on both sides.
                if (X.h * X.w != 1) fail(op + " " + n.name + ": only supported on [N,C,1,1] / [N,C] tensors");
                if (op == "Flatten") {
                    if (on.attr_i("axis", 1) != 1) fail("Flatten " + n.name + ": only axis=1 is supported");
                    odims = {X.n, X.c};
                } else if (op == "Squeeze") odims = {X.n, X.c};
                else {
                    const OnnxTensor* shp = on.inputs.size() > 1 ? L.init(on.inputs[1]) : nullptr;
                    if (!shp) fail("Reshape " + n.name + ": shape must be an initializer");
                    std::vector<int64_t> d = shp->i;
                    int64_t known = 1, neg = -1;
                    for (size_t k = 0; k < d.size(); ++k) {
                        if (d[k] == 0 && k < X.dims.size()) d[k] = X.dims[k];
                        if (d[k] == -1) neg = int64_t(k); else known *= d[k];
                    }
                    if (neg >= 0) d[size_t(neg)] = X.n * X.c / known;
                    int64_t tot = 1; for (auto v : d) tot *= v;
                    if (tot != X.n * X.c || d.empty() || d[0] != X.n) fail("Reshape " + n.name + ": must keep the batch axis");
                    for (size_t k = 2; k < d.size(); ++k) if (d[k] != 1) fail("Reshape " + n.name + ": unsupported target shape");
                    odims = d;
                }
            }
        } else {
            fail("Unsupported ONNX operator: " + op + " (node " + n.name + ")");
        }
        n.out = L.new_val(on.outputs[0], odims);
        L.vals[n.out].producer = int(L.nodes.size());
        L.nodes.push_back(std::move(n));
    }

    for (const auto& vo : m.outputs) {
        int v = L.get_val(vo.name);
        L.vals[v].is_output = true;
    }

    auto single_consumer = [&](int v) { return !L.vals[v].is_output && L.consumers(v).size() == 1; };

    // ---- fusion 1: merge Affine->Affine chains (BN followed by Caffe-style Scale Mul/Add) ----------
    for (size_t i = 0; i < L.nodes.size(); ++i) {
        LNode& a = L.nodes[i];
        if (a.dead || a.kind != L_AFFINE) continue;
        while (single_consumer(a.out)) {
            int ci = L.consumers(a.out)[0];
            LNode& b = L.nodes[ci];
            if (b.kind != L_AFFINE) break;
            for (size_t c = 0; c < a.s.size(); ++c) {
                a.t[c] = b.s[c] * a.t[c] + b.t[c];
                a.s[c] = b.s[c] * a.s[c];
            }
            a.name += "+" + b.name;
            a.out = b.out;
            L.vals[a.out].producer = int(i);
            b.dead = true;
        }
    }
    // ---- fusion 2: conv epilogues (Conv -> Affine -> Relu) -----------------------------------------
    for (size_t i = 0; i < L.nodes.size(); ++i) {
        LNode& cv = L.nodes[i];
        if (cv.dead || cv.kind != L_CONV) continue;
        int64_t cout = L.vals[cv.out].c;
        size_t kper = cv.w.size() / size_t(cout);
        while (single_consumer(cv.out) && !cv.relu) {
            int ci = L.consumers(cv.out)[0];
            LNode& b = L.nodes[ci];
            if (b.kind == L_AFFINE) {
                // Once a residual has been absorbed the epilogue computes conv + bias + res: scaling weights and bias would leave
                // the shortcut unscaled (pre-activation / ResNet-v2 blocks: Conv -> Add -> BN -> ReLU).  The BN then becomes the
                // consumer's prologue (fusion 3) or a standalone eltwise step (fusion 4).
                if (cv.res >= 0) break;
                if (cv.bias.empty()) cv.bias.assign(size_t(cout), 0.f);
                for (int64_t o = 0; o < cout; ++o) {
                    for (size_t k = 0; k < kper; ++k) cv.w[size_t(o) * kper + k] *= b.s[o];
                    cv.bias[o] = cv.bias[o] * b.s[o] + b.t[o];
                }
            } else if (b.kind == L_RELU) cv.relu = true;
            else if (b.kind == L_ADD && cv.res < 0) {
                // residual shortcut: fold the Add into this conv's epilogue when the other operand already exists at this point of
                // the schedule (its producer runs earlier); otherwise the other branch's conv picks the Add up when its turn comes
                const int other = b.in[0] == cv.out ? b.in[1] : b.in[0];
                const int op = L.vals[other].producer;
                if (other == cv.out || !(L.vals[other].is_input ? false : op >= 0 && op < int(i))) break;
                cv.res = other;
                cv.in.push_back(other);
            } else break;
            cv.name += "+" + b.name;
            cv.out = b.out;
            L.vals[cv.out].producer = int(i);
            b.dead = true;
        }
    }
    // ---- fusion 2c: Conv1x1 -> AveragePool  ==>  AveragePool -> Conv1x1 ---------------------------------
    // Both are linear and a 1x1/stride-1 conv acts per pixel, so they commute (the conv's bias too: the mean of a constant is the
    // constant).  DenseNet's transitions (BN -> ReLU -> Conv1x1 -> AvgPool2x2) then run their conv on a quarter of the pixels:
    // 4x fewer FLOPs for those layers (8 % of the network), and the BN+ReLU prologue rides on the pool.  Only when the pool window
    // tiles the image exactly (no padding, no partial windows), so every output averages the same number of inputs.
    if (!env.get("IE_NO_POOL_SWAP")) {
        for (size_t i = 0; i < L.nodes.size(); ++i) {
            if (L.nodes[i].dead || L.nodes[i].kind != L_CONV) continue;
            const LNode& cv0 = L.nodes[i];
            if (cv0.kh != 1 || cv0.kw != 1 || cv0.sh != 1 || cv0.sw != 1 || cv0.pt || cv0.pl || cv0.pb || cv0.pr || cv0.relu || cv0.res >= 0) continue;
            if (!single_consumer(cv0.out)) continue;
            const int pj = L.consumers(cv0.out)[0];
            const LNode& pl0 = L.nodes[pj];
            const Val& X = L.vals[cv0.in[0]];
            if (pl0.kind != L_AVGPOOL || pl0.pt || pl0.pl || pl0.pb || pl0.pr || pl0.kh != pl0.sh || pl0.kw != pl0.sw || X.h % pl0.kh || X.w % pl0.kw ||
                X.is_input)
                continue;
            // new value: the pooled conv input [N, Cin, H/k, W/k]
            const int x = cv0.in[0], conv_out = cv0.out, pool_out = pl0.out;
            const int pooled = L.new_val(L.vals[x].name + "/pooled@" + pl0.name, {X.n, X.c, X.h / pl0.kh, X.w / pl0.kw});
            LNode conv = L.nodes[i], pool = L.nodes[size_t(pj)];
            pool.in = {x};
            pool.out = pooled;
            pool.name = pl0.name + "(before " + cv0.name + ")";
            conv.in[0] = pooled;
            conv.out = pool_out;                        // same dims as before: [N, Cout, H/k, W/k]
            (void)conv_out;                             // the full-resolution conv output no longer exists
            // the pool takes the conv's slot in the schedule, the conv the pool's (everything in between is independent of both)
            L.nodes[i] = pool;
            L.nodes[size_t(pj)] = conv;
            L.vals[pooled].producer = int(i);
            L.vals[pool_out].producer = pj;
            L.vals[conv_out].producer = -1;
        }
    }
    // ---- fusion 3: prologues (Affine -> Relu -> {Conv, GlobalAveragePool, swapped AveragePool}) ---------
    for (size_t i = 0; i < L.nodes.size(); ++i) {
        LNode& cv = L.nodes[i];
        if (cv.dead || (cv.kind != L_CONV && cv.kind != L_GAP && cv.kind != L_AVGPOOL)) continue;
        int x = cv.in[0];
        bool took_relu = false;
        int p = L.vals[x].producer;
        if (p >= 0 && !L.nodes[p].dead && L.nodes[p].kind == L_RELU && single_consumer(x)) {
            cv.pre_relu = true;
            cv.has_pre = true;
            cv.name = L.nodes[p].name + "+" + cv.name;
            L.nodes[p].dead = true;
            x = L.nodes[p].in[0];
            took_relu = true;
        }
        p = L.vals[x].producer;
        if (p >= 0 && !L.nodes[p].dead && L.nodes[p].kind == L_AFFINE && !L.vals[x].is_output) {
            // live consumers of x: none if its Relu was just absorbed, else exactly this node
            std::vector<int> cons = L.consumers(x);
            bool ok = took_relu ? cons.empty() : (cons.size() == 1 && cons[0] == int(i));
            if (ok) {
                cv.pre_s = L.nodes[p].s;
                cv.pre_t = L.nodes[p].t;
                cv.has_pre = true;
                cv.name = L.nodes[p].name + "+" + cv.name;
                L.nodes[p].dead = true;
                x = L.nodes[p].in[0];
            }
        }
        if (cv.has_pre && cv.pre_s.empty()) {
            cv.pre_s.assign(size_t(L.vals[x].c), 1.f);
            cv.pre_t.assign(size_t(L.vals[x].c), 0.f);
        }
        cv.in[0] = x;
    }
    // ---- fusion 4: Add -> Relu, Affine -> Relu -----------------------------------------------------
    for (size_t i = 0; i < L.nodes.size(); ++i) {
        LNode& a = L.nodes[i];
        if (a.dead || (a.kind != L_ADD && a.kind != L_AFFINE)) continue;
        if (single_consumer(a.out)) {
            LNode& b = L.nodes[L.consumers(a.out)[0]];
            if (b.kind == L_RELU) {
                a.relu = true;
                a.name += "+" + b.name;
                a.out = b.out;
                L.vals[a.out].producer = int(i);
                b.dead = true;
            }
        }
    }

    // ---- graph inputs in NCHW: only convs can read them strided; otherwise stage an NHWC copy ------
    {
        std::vector<LNode> pre;
        for (size_t v = 0; v < L.vals.size(); ++v) {
            if (!L.vals[v].is_input || !L.vals[v].input_nchw) continue;
            bool all_conv = true;
            for (int ci : L.consumers(int(v)))
                if (L.nodes[ci].kind != L_CONV || L.nodes[ci].in[0] != int(v) || L.nodes[ci].res == int(v)) all_conv = false;
            if (all_conv && !L.vals[v].is_output) continue;
            LNode cp;
            cp.kind = L_COPY;
            cp.name = "nchw_to_nhwc(" + L.vals[v].name + ")";
            cp.in = {int(v)};
            cp.out = L.new_val(L.vals[v].name + "/nhwc", L.vals[v].dims);
            for (auto& nd : L.nodes) if (!nd.dead) for (int& x : nd.in) if (x == int(v)) x = cp.out;
            pre.push_back(cp);
        }
        if (!pre.empty()) {
            size_t shift = pre.size();
            L.nodes.insert(L.nodes.begin(), pre.begin(), pre.end());
            for (auto& val : L.vals) if (val.producer >= 0) val.producer += int(shift);
            for (size_t k = 0; k < shift; ++k) L.vals[L.nodes[k].out].producer = int(k);
        }
    }

    // ---- graph outputs must end up dense (NCHW order) in their own buffer --------------------------
    std::vector<int> out_vals;
    for (const auto& vo : m.outputs) {
        int v = L.get_val(vo.name);
        // Always materialise through a copy when the tensor is spatial (NHWC->NCHW) or is a graph input;
        // [N,C,1,1] / [N,C] tensors are already in ABI order and only need their own dense buffer.
        bool spatial = L.vals[v].h * L.vals[v].w > 1;
        bool consumed = !L.consumers(v).empty();
        if (spatial || L.vals[v].is_input || consumed) {
            LNode cp;
            cp.kind = L_COPY;
            cp.name = "to_output(" + L.vals[v].name + ")";
            cp.in = {v};
            cp.out = L.new_val(L.vals[v].name + "/out", L.vals[v].dims);
            L.vals[cp.out].is_output = true;
            L.vals[cp.out].input_nchw = spatial;   // reused flag: dense NCHW layout
            L.vals[v].is_output = false;
            L.vals[cp.out].producer = int(L.nodes.size());
            L.nodes.push_back(cp);
            out_vals.push_back(cp.out);
        } else out_vals.push_back(v);
    }

    // ---- concat / alias placement ---------------------------------------------------------------
    for (int i = int(L.nodes.size()) - 1; i >= 0; --i) {
        LNode& n = L.nodes[i];
        if (n.dead) continue;
        if (n.kind == L_ALIAS) {
            Val& src = L.vals[n.in[0]];
            // the alias output shares storage with its input: make the *input* a child of the output
            if (src.parent < 0 && !src.is_input && !src.is_output) { src.parent = n.out; src.parent_off = 0; }
            else {  // cannot alias: degrade to a copy
                n.kind = L_COPY;
            }
        } else if (n.kind == L_CONCAT) {
            int64_t off = 0;
            for (size_t k = 0; k < n.in.size(); ++k) {
                Val& src = L.vals[n.in[k]];
                bool dup = false;
                for (size_t j = 0; j < k; ++j) if (n.in[j] == n.in[k]) dup = true;
                if (src.parent < 0 && !src.is_input && !src.is_output && !dup) { src.parent = n.out; src.parent_off = off; }
                off += src.c;
            }
        }
    }
    // roots, absolute offsets
    std::vector<int64_t> pitch(L.vals.size(), 0);
    for (size_t v = 0; v < L.vals.size(); ++v) {
        int r = int(v);
        int64_t off = 0;
        while (L.vals[r].parent >= 0) { off += L.vals[r].parent_off; r = L.vals[r].parent; }
        L.vals[v].root = r;
        L.vals[v].abs_off = off;
    }

    // ---- liveness over live nodes, buffer recycling -------------------------------------------------
    std::vector<int> order;
    for (size_t i = 0; i < L.nodes.size(); ++i) if (!L.nodes[i].dead) order.push_back(int(i));
    const int INF = 1 << 30;
    std::vector<int> first_def(L.vals.size(), INF), last_use(L.vals.size(), -1);
    std::vector<char> used(L.vals.size(), 0);
    for (size_t v = 0; v < L.vals.size(); ++v)
        if (L.vals[v].is_input) { first_def[L.vals[v].root] = -1; used[L.vals[v].root] = 1; }
    for (size_t pos = 0; pos < order.size(); ++pos) {
        const LNode& n = L.nodes[order[pos]];
        int r = L.vals[n.out].root;
        used[r] = 1;
        first_def[r] = std::min(first_def[r], int(pos));
        last_use[r] = std::max(last_use[r], int(pos));
        for (int x : n.in) { int rx = L.vals[x].root; last_use[rx] = std::max(last_use[rx], int(pos)); used[rx] = 1; }
    }
    // Dense fusion (see the pass after step emission): when a 3x3 conv and the 1x1 conv behind it will run as ONE launch, the 3x3's
    // input (the bottleneck tensor, read as halo by neighbouring tiles) must outlive the launch that also writes the next
    // bottleneck: keep it live one position longer so the two never share a buffer.
    if (precision == Precision::F32 && !env.get("IE_NO_DENSE_FUSE"))
        for (size_t pos = 0; pos + 1 < order.size(); ++pos) {
            const LNode& a3 = L.nodes[order[pos]];
            size_t nxt = pos + 1;                      // Concat / alias nodes emit nothing: the launch behind the 3x3 is the next real node
            while (nxt < order.size() && (L.nodes[order[nxt]].kind == L_CONCAT || L.nodes[order[nxt]].kind == L_ALIAS)) ++nxt;
            if (nxt >= order.size()) break;
            const LNode& b1 = L.nodes[order[nxt]];
            if (a3.kind != L_CONV || b1.kind != L_CONV || a3.kh != 3 || a3.kw != 3 || b1.kh != 1 || b1.kw != 1 || a3.has_pre) continue;
            if (L.vals[a3.out].c != 32 || L.vals[b1.out].c != 128 || L.vals[b1.in[0]].root != L.vals[a3.out].root) continue;
            if (L.vals[a3.out].n * L.vals[a3.out].h * L.vals[a3.out].w > FuseMaxPixels(env)) continue;
            const int rb = L.vals[a3.in[0]].root;
            last_use[rb] = std::max(last_use[rb], int(nxt));
        }
    for (int v : out_vals) last_use[L.vals[v].root] = INF;
    for (size_t v = 0; v < L.vals.size(); ++v) if (L.vals[v].is_input) last_use[L.vals[v].root] = INF;  // staging buffers stay dedicated

    auto root_floats = [&](int r) {
        const Val& R = L.vals[r];
        return R.n * R.c * R.h * R.w;
    };
    std::multimap<int64_t, int> free_pool;   // size -> buffer id
    auto alloc_buf = [&](int r) {
        int64_t need = root_floats(r);
        // fp8 mode types buffers by tensor shape ([N, C] vectors are halfs, spatial tensors e4m3): vectors get their own buffers so
        // a recycled buffer never changes element type
        bool dedicated = last_use[r] == INF || (precision == Precision::F8 && L.vals[r].h * L.vals[r].w == 1);
        L.vals[r].dedicated = dedicated;
        if (!dedicated) {
            auto it = free_pool.lower_bound(need);
            // accept a recycled buffer up to 2x the needed size; otherwise grow a new one
            if (it != free_pool.end() && it->first <= 2 * need) {
                int b = it->second;
                free_pool.erase(it);
                L.vals[r].buf = b;
                return;
            }
        }
        plan.buffer_floats.push_back(need);
        L.vals[r].buf = int(plan.buffer_floats.size()) - 1;
    };
    for (size_t v = 0; v < L.vals.size(); ++v)
        if (used[v] && L.vals[v].root == int(v) && first_def[v] == -1) alloc_buf(int(v));
    for (size_t pos = 0; pos < order.size(); ++pos) {
        for (size_t v = 0; v < L.vals.size(); ++v)
            if (used[v] && L.vals[v].root == int(v) && first_def[v] == int(pos)) alloc_buf(int(v));
        for (size_t v = 0; v < L.vals.size(); ++v)
            if (used[v] && L.vals[v].root == int(v) && last_use[v] == int(pos) && L.vals[v].buf >= 0 && !L.vals[v].dedicated)
                free_pool.insert({plan.buffer_floats[size_t(L.vals[v].buf)], L.vals[v].buf});
    }

    // fp16 mode: every buffer except the graph's own inputs/outputs (dedicated, never recycled) holds halfs
    auto mark_buffer_types = [&] {
        plan.buffer_f16.assign(plan.buffer_floats.size(), precision == Precision::F16 ? 1 : (precision == Precision::F8 ? 2 : 0));
        if (precision == Precision::F8)
            for (size_t v = 0; v < L.vals.size(); ++v)
                if (used[v] && L.vals[v].root == int(v) && L.vals[v].buf >= 0 && L.vals[v].h * L.vals[v].w == 1) plan.buffer_f16[size_t(L.vals[v].buf)] = 1;
        for (size_t v = 0; v < L.vals.size(); ++v)
            if (L.vals[v].is_input && L.vals[L.vals[v].root].buf >= 0) plan.buffer_f16[size_t(L.vals[L.vals[v].root].buf)] = 0;
        for (int v : out_vals)
            if (L.vals[L.vals[v].root].buf >= 0) plan.buffer_f16[size_t(L.vals[L.vals[v].root].buf)] = 0;
    };
    mark_buffer_types();

    auto view_of = [&](int v) {
        const Val& X = L.vals[v];
        const Val& R = L.vals[X.root];
        View w;
        w.buf = R.buf;
        w.f16 = R.buf >= 0 && plan.buffer_f16[size_t(R.buf)] == 1;
        w.f8 = R.buf >= 0 && plan.buffer_f16[size_t(R.buf)] == 2;
        w.n = X.n; w.c = X.c; w.h = X.h; w.w = X.w;
        w.c_off = X.abs_off;
        w.pitch = R.c;
        w.nchw = (X.is_input || X.is_output) && X.input_nchw;
        if (w.buf < 0) fail("internal planner error: value " + X.name + " has no buffer");
        return w;
    };

    // ---- emit steps --------------------------------------------------------------------------------
    auto push_vec = [&](const std::vector<float>& v) {
        while (plan.weights.size() % 8) plan.weights.push_back(0.f);   // 32 B in the fp32 blob, 16 B in its half mirror
        int64_t off = int64_t(plan.weights.size());
        plan.weights.insert(plan.weights.end(), v.begin(), v.end());
        return off;
    };
    auto vbytes = [](const View& v) { return double(v.numel()) * double(v.esize()); };
    // who wrote what: (buffer, channel offset, channels) -> step index, for Step::in_src / in2_src
    std::map<std::vector<int64_t>, int> writer;
    auto src_of = [&](const View& v) {
        auto it = writer.find({int64_t(v.buf), v.c_off, v.c});
        return it == writer.end() ? -1 : it->second;
    };
    for (int idx : order) {
        const LNode& n = L.nodes[idx];
        if (n.kind == L_ALIAS) continue;
        Step s;
        s.name = n.name;
        if (n.kind == L_CONCAT) {
            // members that could not be placed in the parent buffer are copied into their slice
            int64_t off = 0;
            for (size_t k = 0; k < n.in.size(); ++k) {
                const Val& src = L.vals[n.in[k]];
                bool placed = src.root == L.vals[n.out].root && src.abs_off == L.vals[n.out].abs_off + off;
                if (!placed) {
                    Step c;
                    c.kind = StepKind::Copy;
                    c.name = n.name + "/copy" + std::to_string(k);
                    c.in = view_of(n.in[k]);
                    c.out = view_of(n.out);
                    c.out.c = src.c;
                    c.out.c_off += off;
                    c.bytes = vbytes(c.in) + vbytes(c.out);
                    if (c.in.f8 || c.out.f8) fail("fp8 precision: concat copy " + c.name + " of an fp8 tensor is not supported");
                    c.idx = int(plan.steps.size());
                    c.in_src = src_of(c.in);
                    writer[{int64_t(c.out.buf), c.out.c_off, c.out.c}] = c.idx;
                    plan.steps.push_back(c);
                }
                off += src.c;
            }
            continue;
        }
        s.in = view_of(n.in[0]);
        s.out = view_of(n.out);
        s.relu = n.relu;
        if (n.has_pre) {
            s.pre_scale_off = push_vec(n.pre_s);
            s.pre_shift_off = push_vec(n.pre_t);
            s.pre_relu = n.pre_relu;
        }
        switch (n.kind) {
            case L_CONV: {
                s.kind = StepKind::Conv;
                s.kh = n.kh; s.kw = n.kw; s.sh = n.sh; s.sw = n.sw; s.pt = n.pt; s.pl = n.pl; s.pb = n.pb; s.pr = n.pr;
                s.w_off = push_vec(n.w);
                if (!n.bias.empty()) s.bias_off = push_vec(n.bias);
                if (n.res >= 0) { s.in2 = view_of(n.res); s.has_in2 = true; }
                int64_t M = s.out.n * s.out.h * s.out.w, N = s.out.c, K = int64_t(n.kh) * n.kw * s.in.c;
                s.flops = 2.0 * double(M) * double(N) * double(K);
                const bool in16 = s.in.f16;
                const bool in8 = s.in.f8;
                s.bytes = vbytes(s.in) + vbytes(s.out) + (in8 ? 1.0 : (in16 ? 2.0 : 4.0)) * double(n.w.size()) + (n.res >= 0 ? vbytes(s.in2) : 0.0);
                bool vec_ok = !in16 && !in8 && !s.in.nchw && s.in.c % 4 == 0 && s.in.pitch % 4 == 0 && s.in.c_off % 4 == 0 && n.kh * n.kw <= 32 &&
                              s.in.n * s.in.h * s.in.w * s.in.pitch * 4 < (int64_t(1) << 31) && int64_t(n.w.size()) * 4 < (int64_t(1) << 31);
                // fp16 MFMA path: 16-byte chunks of 8 halfs, so channel counts / slice offsets must be multiples of 8
                const bool vec16_ok = in16 && !s.in.nchw && s.in.c % 8 == 0 && s.in.pitch % 8 == 0 && s.in.c_off % 8 == 0 && n.kh * n.kw <= 32 &&
                                      s.in.n * s.in.h * s.in.w * s.in.pitch * 2 < (int64_t(1) << 31) && int64_t(n.w.size()) * 2 < (int64_t(1) << 31) &&
                                      s.out.n * s.out.h * s.out.w * s.out.pitch < (int64_t(1) << 31);
                const int64_t bk = in16 ? 2 * kIgemmBK : kIgemmBK;       // K-tile depth in elements (128 B per LDS row either way)
                // one-thread-per-output only for toy problems (test_model's 3->5->2 MLP): at batch 1 DenseNet's block-4 convs have
                // M*N = 1568 outputs but K = 1152 - the naive kernel took 141 us there, the MFMA kernels 13 us
                if (M * N * K <= 32768 || (M * N < 2048 && K <= 4096 && !(vec_ok || vec16_ok))) s.algo = ConvAlgo::Naive;
                else if (vec_ok || vec16_ok) s.algo = ConvAlgo::IgemmVec;
                else if (K <= 2048 && !in16) s.algo = ConvAlgo::IgemmScalar;
                else s.algo = ConvAlgo::Naive;
                // the stem of an image classifier (7x7 / stride 2 / pad 3 over the 3-channel NCHW graph input) has its own kernel
                const bool stem_ok = s.in.nchw && !in16 && s.in.c == 3 && n.kh == 7 && n.kw == 7 && n.sh == 2 && n.sw == 2 && n.pt == 3 &&
                                     n.pl == 3 && n.pb == 3 && n.pr == 3 && !n.has_pre && N <= 64 && N % 8 == 0 && s.out.pitch % 8 == 0 &&
                                     s.out.c_off % 8 == 0 && s.in.numel() * 4 < (int64_t(1) << 31) &&
                                     s.out.n * s.out.h * s.out.w * s.out.pitch * 4 < (int64_t(1) << 31);
                if (stem_ok && M * N >= 2048 && !(s.out.f8 && (N % 16 || s.out.pitch % 16 || s.out.c_off % 16))) s.algo = ConvAlgo::Stem;
                if (in8 || s.out.f8) {
                    // fp8 mode: e4m3 tensors are only understood by the fp8 kernels; anything they cannot run is a load error, never a
                    // silent reinterpretation of the bytes by another kernel
                    if (s.out.f8 && !in8) {
                        if (s.algo != ConvAlgo::Stem)
                            fail("fp8 precision: conv " + n.name + " reads a non-fp8 tensor and writes an fp8 one; only the 7x7/s2 stem over the fp32 graph input does that");
                    } else {
                        if (n.has_pre) fail("fp8 precision: conv " + n.name + " has an activation prologue (pre-activation graphs are not supported in fp8 mode)");
                        if (!s.out.f8) fail("fp8 precision: conv " + n.name + " reads an fp8 tensor and writes a non-fp8 one");
                        if (s.in.nchw || s.in.c % 16 || s.in.pitch % 16 || s.in.c_off % 16 || N % 16 || s.out.pitch % 16 || s.out.c_off % 16 || n.kh * n.kw > 32)
                            fail("fp8 precision: conv " + n.name + " needs channel counts and slice offsets that are multiples of 16");
                        if (n.res >= 0 && !s.in2.f8) fail("fp8 precision: the shortcut of conv " + n.name + " is not an fp8 tensor");
                        s.algo = ConvAlgo::IgemmF8;
                    }
                }
                s.tile = choose_tile(M, N);
                s.splitk = 1;
                const int heuristic_tile = s.tile;
                s.base_tile = heuristic_tile;
                // ---- plan-time eligibility of the specialised kernels (the launchers re-check pointers / alignment; the executor falls
                //      back to the tiled implicit GEMM when a launcher declines) ----
                const bool is1x1 = n.kh == 1 && n.kw == 1 && n.sh == 1 && n.sw == 1 && n.pt == 0 && n.pl == 0 && n.pb == 0 && n.pr == 0;
                const bool is3x3 = n.kh == 3 && n.kw == 3 && n.sh == 1 && n.sw == 1 && n.pt == 1 && n.pl == 1 && n.pb == 1 && n.pr == 1;
                static const int ws_tn[14] = {4, 4, 2, 2, 1, 1, 4, 4, 2, 2, 1, 1, 1, 1};      // 12, 13: fp32 K-split variants (8 / 4 waves)
                auto ws16_ok = [&](int t) {
                    if (t < 0 || t >= 18) return false;          // (12-17: the one-workgroup-per-CU grids of shapes 0-5)
                    const int tn = ws_tn[t % 6];
                    return vec16_ok && is1x1 && s.in.c % 32 == 0 &&
                           (32 * tn * (s.in.c + 8) + 2 * s.in.c) * 2 + 128 * tn <= 160 * 1024 && !(tn > 1 && N <= 32 * (tn / 2)) &&
                           (s.out.f16 ? (N % 8 == 0 && s.out.pitch % 8 == 0 && s.out.c_off % 8 == 0) : (N % 4 == 0 && s.out.pitch % 4 == 0 && s.out.c_off % 4 == 0));
                };
                auto ws32_ok = [&](int t) {
                    if (t >= 14 && t < 20) t -= 14;              // 14-19: shapes 0-5 on a grid of one workgroup per CU: same operand conditions
                    return t >= 0 && t < 14 && vec_ok && !s.out.f16 && is1x1 && s.in.c % 16 == 0 &&
                           (t < 12 ? (32 * ws_tn[t] * (s.in.c + 4) + 2 * s.in.c + 32 * ws_tn[t]) * 4 <= 160 * 1024
                                   : (s.in.c / 16 >= (t == 12 ? 8 : 4) &&
                                      (32 * (s.in.c + 4) + 2 * s.in.c + 32 + (t == 12 ? 4 : 2) * 32 * 36) * 4 <= 160 * 1024)) &&
                           !(ws_tn[t] > 1 && N <= 32 * (ws_tn[t] / 2)) && N % 4 == 0 && s.out.pitch % 4 == 0 && s.out.c_off % 4 == 0;
                };
                static const int ws3_cfg[5][3] = {{4, 2, 12}, {4, 1, 8}, {8, 1, 6}, {2, 1, 12}, {12, 1, 6}};   // waves, row blocks per wave, prefetch depth (kWs3Tiles, kernels_ws.hip)
                auto ws3_ok = [&](int t3) {
                    if (t3 < 0 || t3 >= 5) return false;
                    const int64_t pr3 = 32 * ws3_cfg[t3][1] * ws3_cfg[t3][0] + 2 * (s.in.w + 1) + 2;
                    return vec16_ok && s.out.f16 && is3x3 && !n.has_pre && N % 8 == 0 && s.out.pitch % 8 == 0 && s.out.c_off % 8 == 0 &&
                           pr3 <= ws3_cfg[t3][2] * (64 * ws3_cfg[t3][0] / 8) && (9 * ((s.in.c + 63) / 64) * 32 + pr3) * 144 + 128 <= 160 * 1024;
                };
                static const int dcfg[6][3] = {{1, 8, 8}, {1, 16, 4}, {1, 9, 8}, {1, 4, 8}, {1, 12, 6}, {2, 8, 4}};   // tn, waves, max chunks
                static const int wcfg[4][3] = {{1, 8, 9}, {2, 8, 9}, {1, 4, 18}, {2, 4, 18}};                      // window variants: tn, waves, max chunks
                auto direct_ok = [&](int t) {
                    if (t < 0 || t >= 15) return false;
                    if (t >= 10) {     // activations-stationary 1x1 (fp32): the workgroup's 32 / 16 pixel rows in LDS, weights streamed from the mirror
                        static const int acfg[5] = {128, 64, 256, 64, 64};                                        // output channels per workgroup
                        return vec_ok && !in16 && !s.out.f16 && !s.has_in2 && is1x1 && s.in.c % 16 == 0 && N % acfg[t - 10] == 0 && s.out.pitch % 4 == 0 &&
                               s.out.c_off % 4 == 0 && M <= (int64_t(1) << 22) && 32 * (s.in.c + 4) * 4 <= 160 * 1024;
                    }
                    if (t >= 6) {      // fp32, output grid == input grid, activations through an LDS window, fragment-major weights
                        const int* wc = wcfg[t - 6];
                        const int64_t total = s.in.c % 16 == 0 ? int64_t(n.kh) * n.kw * (s.in.c / 16) : 0;
                        const int64_t win = (16 + (n.kh - 1) * s.in.w + (n.kw - 1)) * (s.in.c + 4) * 4, part = int64_t(wc[1]) * 16 * (16 * wc[0] + 4) * 4;
                        return vec_ok && !in16 && !s.out.f16 && !s.has_in2 && s.in.c % 16 == 0 && N % (16 * wc[0]) == 0 && n.sh == 1 && n.sw == 1 &&
                               s.out.h == s.in.h && s.out.w == s.in.w && n.pt < n.kh && n.pl < n.kw && s.out.pitch % 2 == 0 && s.out.c_off % 2 == 0 &&
                               total >= wc[1] && total <= wc[1] * wc[2] && M <= 65536 && n.kh * n.kw <= 49 && std::max(win, part) <= 160 * 1024;
                    }
                    const int cw = in16 ? 32 : 16, al = in16 ? 8 : 4;
                    const int64_t total = s.in.c % cw == 0 ? int64_t(n.kh) * n.kw * (s.in.c / cw) : 0;
                    return (vec_ok || vec16_ok) && s.in.c % cw == 0 && s.in.pitch % al == 0 && s.in.c_off % al == 0 && N % 2 == 0 && s.out.pitch % 2 == 0 &&
                           s.out.c_off % 2 == 0 && total >= dcfg[t][1] && total <= dcfg[t][1] * dcfg[t][2] && !(dcfg[t][0] > 1 && N <= 32) && M <= 65536 &&
                           n.kh * n.kw <= 49;
                };
                const bool raster_ok = vec_ok && !s.out.f16 && is3x3 && !n.has_pre;
                const bool wino_ok = raster_ok && N == 32 && s.in.c % 32 == 0 && s.in.h % 2 == 0 && s.in.w % 2 == 0 && n.res < 0 && s.out.pitch % 4 == 0 && s.out.c_off % 4 == 0;
                // ---- default choice without the autotuner (IE_AUTOTUNE=0, or before Prepare() has timed anything): the kernels
                //      the exhaustive search picks for DenseNet / ResNet shapes ----
                if (s.algo == ConvAlgo::IgemmF8) {
                    if (const char* ft = env.get("IE_FORCE_TILE")) {
                        const int t = std::atoi(ft);
                        if (t >= 0 && t < kNumIgemmBaseTiles && !(kIgemmTiles[t].bn > 32 && N <= 32)) s.tile = t;
                        // >= 100: the weights-stationary 1x1 kernel's tiles, >= 200: the 3x3's (kernels_ws8.hip); a launcher that declines the
                        // operands hands the step back to the tiled kernel (executor)
                        if (t >= 100 && t < 110 && n.kh == 1 && n.kw == 1) s.tile = t;           // (strided 1x1 convs too: the kernel's STR form)
                        if (t >= 200 && t < 208 && is3x3) s.tile = t;
                    }
                    break;
                }
                if (!env.get("IE_FORCE_ALGO") && !env.get("IE_FORCE_TILE") && s.algo == ConvAlgo::IgemmVec) {
                    int pick = -1;
                    if (M <= 2048) {                                                       // tiny grids: split K over the waves
                        if (is3x3 && direct_ok(6)) pick = 6;                             // 16-pixel window tiles: 4x the workgroups
                        else if (is1x1 && direct_ok(13)) pick = 13;                      // 16-pixel activations-stationary tiles
                        for (int t : {1, 0, 4, 3}) if (pick < 0 && direct_ok(t)) pick = t;
                        if (pick >= 0) { s.algo = ConvAlgo::Direct; s.tile = pick; }
                    } else if (in16) {
                        if (is1x1) { for (int t : {0, 2, 4}) if (pick < 0 && ws16_ok(t)) pick = t; if (pick >= 0) { s.algo = ConvAlgo::Ws1x1; s.tile = pick; } }
                        else if (is3x3) { for (int t : {2, 1, 0}) if (pick < 0 && ws3_ok(t)) pick = t; if (pick >= 0) { s.algo = ConvAlgo::Ws3x3; s.tile = pick; } }
                    } else {
                        if (is1x1 && direct_ok(10)) { s.algo = ConvAlgo::Direct; s.tile = 10; }      // activations-stationary 1x1: 128-channel multiples whose 32 pixel rows fit in LDS
                        else if (is1x1 && M >= 20000) { for (int t : {0, 2, 4}) if (pick < 0 && ws32_ok(t)) pick = t; if (pick >= 0) { s.algo = ConvAlgo::Ws1x1; s.tile = pick; } }
                        else if (wino_ok && M >= 20000) { s.algo = ConvAlgo::Wino3x3; s.tile = 5; }     // Winograd F(2x2,3x3), 2x14 tiles, eight waves (falls back to the tiled kernel without the U mirror)
                        else if (raster_ok && M >= 20000 && N <= 64) { s.algo = ConvAlgo::Raster3x3; s.tile = N <= 32 ? 0 : 4; }
                        else if (is3x3 && M <= 8192) { if (direct_ok(4)) { s.algo = ConvAlgo::Direct; s.tile = 4; } }
                    }
                }
                // Test / tuning overrides (read at plan time): IE_FORCE_TILE=<n>, IE_FORCE_ALGO=naive|scalar|igemm|raster|ws|direct
                if (const char* fa = env.get("IE_FORCE_ALGO")) {
                    std::string f = fa;
                    if (f == "naive") s.algo = ConvAlgo::Naive;
                    else if (f == "scalar" && K <= 2048 && !in16) s.algo = ConvAlgo::IgemmScalar;
                    else if (f == "igemm" && s.algo == ConvAlgo::Stem) s.algo = K <= 2048 ? ConvAlgo::IgemmScalar : ConvAlgo::Naive;
                    else if (f == "igemm" && s.algo == ConvAlgo::Naive)
                        s.algo = (vec_ok || vec16_ok) ? ConvAlgo::IgemmVec : (K <= 2048 && !in16 ? ConvAlgo::IgemmScalar : ConvAlgo::Naive);
                    else if (f == "ws") {
                        int t = 0;
                        if (const char* ft = env.get("IE_FORCE_TILE")) { int v = std::atoi(ft); if (v >= 0 && v < (in16 ? 18 : 20)) t = v; }
                        if (ws16_ok(t) || ws32_ok(t)) { s.algo = ConvAlgo::Ws1x1; s.tile = t; }
                        else if (ws3_ok(t % 5)) { s.algo = ConvAlgo::Ws3x3; s.tile = t % 5; }
                        else if (s.algo == ConvAlgo::Naive && vec16_ok) s.algo = ConvAlgo::IgemmVec;
                    }
                    else if (f == "direct") {
                        int t = 0;
                        if (const char* ft = env.get("IE_FORCE_TILE")) { int v = std::atoi(ft); if (v >= 0 && v < 15) t = v; }
                        if (direct_ok(t)) { s.algo = ConvAlgo::Direct; s.tile = t; }
                        else if (s.algo == ConvAlgo::Naive && (vec_ok || vec16_ok)) s.algo = ConvAlgo::IgemmVec;
                    }
                    else if (f == "x6") {
                        const bool x6_ok = vec_ok && !s.out.f16 && is1x1 && n.sh == 1 && n.sw == 1 && s.in.c % 32 == 0 && N % 128 == 0 && n.res < 0;
                        if (x6_ok) {
                            s.algo = ConvAlgo::X6;
                            s.tile = 0;
                            if (const char* ft = env.get("IE_FORCE_TILE")) { int t = std::atoi(ft); if (t >= 0 && t < 2) s.tile = t; }
                        }
                    }
                    else if (f == "wino") {
                        if (wino_ok) {
                            s.algo = ConvAlgo::Wino3x3;
                            s.tile = 0;
                            if (const char* ft = env.get("IE_FORCE_TILE")) { int t = std::atoi(ft); if (t >= 0 && t < 12) s.tile = t; }
                        } else if (s.algo == ConvAlgo::Naive)
                            s.algo = (vec_ok || vec16_ok) ? ConvAlgo::IgemmVec : (K <= 2048 && !in16 ? ConvAlgo::IgemmScalar : ConvAlgo::Naive);
                    }
                    else if (f == "raster") {
                        if (raster_ok) {
                            s.algo = ConvAlgo::Raster3x3;
                            s.tile = 0;
                            if (const char* ft = env.get("IE_FORCE_TILE")) { int t = std::atoi(ft); if (t >= 0 && t < 8) s.tile = t; }
                        } else if (s.algo == ConvAlgo::Naive)
                            s.algo = (vec_ok || vec16_ok) ? ConvAlgo::IgemmVec : (K <= 2048 && !in16 ? ConvAlgo::IgemmScalar : ConvAlgo::Naive);
                    }
                }
                if (const char* ft = (s.algo == ConvAlgo::Raster3x3 || s.algo == ConvAlgo::Ws1x1 || s.algo == ConvAlgo::Ws3x3 || s.algo == ConvAlgo::Stem || s.algo == ConvAlgo::Direct || s.algo == ConvAlgo::Wino3x3 || s.algo == ConvAlgo::X6) ? nullptr
                                                                                                                                          : env.get("IE_FORCE_TILE")) {
                    int t = std::atoi(ft);
                    if (t >= 0 && t < kNumIgemmTiles && (t < kNumIgemmBaseTiles || s.algo == ConvAlgo::IgemmVec) && !(in16 && kIgemmTiles[t].deep)) s.tile = t;
                }
                if (s.algo != ConvAlgo::IgemmVec && s.algo != ConvAlgo::Raster3x3 && s.algo != ConvAlgo::Ws1x1 && s.algo != ConvAlgo::Ws3x3 && s.algo != ConvAlgo::Direct && s.algo != ConvAlgo::Wino3x3 && s.algo != ConvAlgo::X6 && s.tile >= kNumIgemmBaseTiles)
                    s.tile = heuristic_tile;       // K-group tiles exist for the vector path only
                if (s.algo == ConvAlgo::Raster3x3) {
                    if (const char* fs = env.get("IE_FORCE_SPLITK")) {
                        int v = std::atoi(fs);
                        if (v >= 1 && v <= 64) s.splitk = v;
                    }
                    if (s.splitk > 1) plan.workspace_floats = std::max<int64_t>(plan.workspace_floats, int64_t(s.splitk) * M * N);
                } else if (s.algo == ConvAlgo::Ws1x1 || s.algo == ConvAlgo::Ws3x3 || s.algo == ConvAlgo::Stem || s.algo == ConvAlgo::Direct || s.algo == ConvAlgo::Wino3x3 || s.algo == ConvAlgo::X6) {
                    s.splitk = 1;
                } else if (s.algo != ConvAlgo::Naive) {
                    // split-K when the output grid cannot fill the chip: aim for >= ~768 workgroups, keep >= 2 K-tiles
                    // per split.  (Deterministic two-pass reduction, see kernels.hip.)
                    const IgemmTile& T = kIgemmTiles[s.tile];
                    const int64_t wgs = ((M + T.bm - 1) / T.bm) * ((N + T.bn - 1) / T.bn);
                    const int64_t cblocks = (s.in.c + bk - 1) / bk;
                    const int64_t KT = s.algo == ConvAlgo::IgemmVec ? int64_t(n.kh) * n.kw * cblocks : (K + kIgemmBK - 1) / kIgemmBK;
                    if (wgs < 384 && KT >= 4) {
                        int64_t want = (768 + wgs - 1) / wgs;
                        s.splitk = int(std::max<int64_t>(1, std::min<int64_t>({want, KT / 2, 32})));
                    }
                    if (const char* fs = env.get("IE_FORCE_SPLITK")) {
                        int v = std::atoi(fs);
                        if (v >= 1 && v <= 64) s.splitk = v;
                    }
                    if (s.splitk > 1) plan.workspace_floats = std::max<int64_t>(plan.workspace_floats, int64_t(s.splitk) * M * N);
                }
                break;
            }
            case L_MAXPOOL: case L_AVGPOOL:
                if ((s.in.f8 || s.out.f8) && (!s.in.f8 || !s.out.f8 || n.has_pre || s.in.c % 16 || s.in.pitch % 16 || s.in.c_off % 16 || s.out.pitch % 16 || s.out.c_off % 16))
                    fail("fp8 precision: pool " + n.name + " needs fp8 operands without a prologue and channel counts that are multiples of 16");
                s.kind = StepKind::Pool;
                s.pool_max = n.kind == L_MAXPOOL;
                s.count_include_pad = n.count_include_pad;
                s.kh = n.kh; s.kw = n.kw; s.sh = n.sh; s.sw = n.sw; s.pt = n.pt; s.pl = n.pl; s.pb = n.pb; s.pr = n.pr;
                s.bytes = vbytes(s.in) + vbytes(s.out);
                s.flops = double(s.out.numel()) * n.kh * n.kw;
                break;
            case L_GAP:
                if (s.in.f8 && (n.has_pre || s.out.f8)) fail("fp8 precision: global pool " + n.name + " with a prologue is not supported");
                s.kind = StepKind::GlobalAvgPool;
                s.bytes = vbytes(s.in) + vbytes(s.out);
                s.flops = double(s.in.numel());
                break;
            case L_AFFINE:
                if (s.in.f8 || s.out.f8) fail("fp8 precision: stand-alone scale/shift " + n.name + " on an fp8 tensor is not supported");
                s.kind = StepKind::Eltwise;
                s.pre_scale_off = push_vec(n.s);
                s.pre_shift_off = push_vec(n.t);
                s.bytes = vbytes(s.in) + vbytes(s.out);
                s.flops = 2.0 * double(s.in.numel());
                break;
            case L_RELU:
                if (s.in.f8 || s.out.f8) fail("fp8 precision: stand-alone Relu " + n.name + " on an fp8 tensor is not supported");
                s.kind = StepKind::Eltwise;
                s.relu = true;
                s.bytes = vbytes(s.in) + vbytes(s.out);
                break;
            case L_ADD:
                if (s.in.f8 || s.out.f8) fail("fp8 precision: stand-alone Add " + n.name + " on fp8 tensors is not supported (only shortcuts folded into a conv)");
                s.kind = StepKind::Eltwise;
                s.in2 = view_of(n.in[1]);
                s.has_in2 = true;
                s.bytes = vbytes(s.in) + vbytes(s.in2) + vbytes(s.out);
                s.flops = double(s.in.numel());
                break;
            case L_COPY:
                if (s.in.f8 || s.out.f8) fail("fp8 precision: layout copy " + n.name + " of an fp8 tensor is not supported");
                s.kind = StepKind::Copy;
                s.bytes = vbytes(s.in) + vbytes(s.out);
                break;
            default: fail("internal planner error: unexpected node kind");
        }
        if (s.kind != StepKind::Conv && s.kind != StepKind::Copy && s.in.nchw)
            fail("internal planner error: NCHW view reached a non-conv step");
        s.idx = int(plan.steps.size());
        s.in_src = src_of(s.in);
        if (s.has_in2) s.in2_src = src_of(s.in2);
        writer[{int64_t(s.out.buf), s.out.c_off, s.out.c}] = int(plan.steps.size());
        plan.total_flops += s.flops;
        plan.total_bytes += s.bytes;
        plan.steps.push_back(std::move(s));
    }
    while (plan.weights.size() % 8) plan.weights.push_back(0.f);

    // ---- dense fusion (fp32, small output grids): 3x3 growth conv of layer L + the 1x1 bottleneck conv of layer L+1 ------------------
    // Pattern: step i = plain 3x3/s1/p1 conv (no prologue, 32 output channels) writing a channel slice of a concat buffer; step i+1 =
    // 1x1 conv with 128 output channels whose input view is that buffer's channels [c_off, slice end): its last 32 input channels are
    // exactly what step i produces, for the same pixels (kernels_fused.hip).  Only where launches are latency-bound (M <= 8192).
    if (precision == Precision::F32 && !env.get("IE_NO_DENSE_FUSE") && !env.get("IE_FORCE_ALGO") && !env.get("IE_FORCE_TILE")) {
        std::vector<Step> fusedsteps;
        for (size_t i = 0; i < plan.steps.size(); ++i) {
            const Step& s3 = plan.steps[i];
            bool fuse = false;
            if (i + 1 < plan.steps.size()) {
                const Step& s1 = plan.steps[i + 1];
                const int64_t M = s3.out.n * s3.out.h * s3.out.w;
                fuse = s3.kind == StepKind::Conv && s1.kind == StepKind::Conv && s3.kh == 3 && s3.kw == 3 && s3.sh == 1 && s3.sw == 1 && s3.pt == 1 && s3.pl == 1 &&
                       s3.pb == 1 && s3.pr == 1 && s3.pre_scale_off < 0 && !s3.has_in2 && s3.out.c == 32 && s3.in.c % 16 == 0 && 9 * (s3.in.c / 16) <= 72 &&
                       9 * (s3.in.c / 16) >= 8 && s1.kh == 1 && s1.kw == 1 && s1.sh == 1 && s1.sw == 1 && s1.pt == 0 && s1.pl == 0 && s1.pb == 0 && s1.pr == 0 &&
                       !s1.has_in2 && s1.out.c == 128 && s1.in.buf == s3.out.buf && s1.in.pitch == s3.out.pitch && !s1.in.nchw && !s3.in.nchw &&
                       s1.in.c_off + s1.in.c == s3.out.c_off + s3.out.c && s1.in.c >= 48 && s1.in.c % 16 == 0 && s1.in.n == s3.out.n && s1.in.h == s3.out.h &&
                       s1.in.w == s3.out.w && s1.out.buf != s3.in.buf && M <= FuseMaxPixels(env) && s3.algo != ConvAlgo::Naive && s1.algo != ConvAlgo::Naive &&
                       s3.in.pitch % 4 == 0 && s3.in.c_off % 4 == 0 && s1.in.pitch % 4 == 0 && s1.in.c_off % 4 == 0 && s1.out.pitch % 4 == 0 && s1.out.c_off % 4 == 0;
                if (fuse) {
                    int pb = M <= 2048 ? 1 : 2;
                    int ftile = 0;
                    if (const char* e = env.get("IE_FUSE_PB")) { const int v = std::atoi(e); if (v == 1 || v == 2) pb = v; if (v == 3) { pb = 1; ftile = 3; } if (v == 4 || v == 5) { pb = v - 3; ftile = v; } }
                    const int64_t px = 16 * pb;
                    const int64_t win = (px + 2 * s3.in.w + 2) * (s3.in.c + 8) * 4, part = 4 * px * 36 * 4;
                    const int64_t c4n = (s1.in.c - 32) / 4, rpp = c4n > 0 && c4n <= 512 ? 512 / c4n : 0;
                    if (std::max(px * (s1.in.c + 8) * 4, win) + part > 160 * 1024 || rpp == 0 || (px + rpp - 1) / rpp > (pb == 1 ? 8 : 16) ||
                        (px + 2 * s3.in.w + 2) * (s3.in.c / 4) > 8 * 512)
                        fuse = false;
                    if (fuse) {
                        Step f = s1;
                        f.algo = ConvAlgo::DenseFused;
                        f.tile = ftile ? ftile : pb;
                        f.splitk = 1;
                        f.name = s3.name + " | " + s1.name;
                        f.flops = s3.flops + s1.flops;
                        f.bytes = s3.bytes + s1.bytes;
                        f.parts = {s3, s1};
                        fusedsteps.push_back(std::move(f));
                        ++i;
                    }
                }
            }
            if (!fuse) fusedsteps.push_back(s3);
        }
        if (fusedsteps.size() != plan.steps.size()) {
            // renumber: idx, in_src / in2_src follow the new positions (a fused step is the producer of both of its outputs)
            std::vector<int> remap(plan.steps.size(), -1);
            for (size_t k = 0; k < fusedsteps.size(); ++k) {
                if (fusedsteps[k].algo == ConvAlgo::DenseFused) { remap[size_t(fusedsteps[k].parts[0].idx)] = int(k); remap[size_t(fusedsteps[k].parts[1].idx)] = int(k); }
                else remap[size_t(fusedsteps[k].idx)] = int(k);
            }
            for (size_t k = 0; k < fusedsteps.size(); ++k) {
                Step& st = fusedsteps[k];
                st.idx = int(k);
                if (st.in_src >= 0) st.in_src = remap[size_t(st.in_src)];
                if (st.in2_src >= 0) st.in2_src = remap[size_t(st.in2_src)];
            }
            plan.steps = std::move(fusedsteps);
        }
    }

    // ---- dense-block chains (fp16, small maps): consecutive dense layers of one concat buffer as ONE step ---------------------------------------
    // Pattern per layer: step i = 1x1/s1 conv with 128 output channels reading channels [c_off, c_off + K) of a concat buffer into a bottleneck
    // tensor T, step i+1 = plain 3x3/s1/p1 conv (no prologue, no residual) from T to 32 channels of the SAME concat buffer outside what the
    // layer reads, T read by nothing else.  A layer touches only its own image, so a workgroup per image walks the whole chain with T in LDS
    // (kernels_block.hip): DenseNet-121 blocks 3-4 at batch 128 go from 80 launches to 2.  Maps of at most 8 x 32 raster positions (14x14, 7x7).
    if (precision == Precision::F16 && !env.get("IE_NO_DENSE_BLOCK") && !env.get("IE_FORCE_ALGO") && !env.get("IE_FORCE_TILE")) {
        auto same_view = [](const View& a, const View& b) {
            return a.buf == b.buf && a.n == b.n && a.c == b.c && a.h == b.h && a.w == b.w && a.c_off == b.c_off && a.pitch == b.pitch && a.nchw == b.nchw && a.f16 == b.f16;
        };
        auto layer_at = [&](size_t i) {
            if (i + 1 >= plan.steps.size()) return false;
            const Step& s1 = plan.steps[i];
            const Step& s3 = plan.steps[i + 1];
            if (s1.kind != StepKind::Conv || s3.kind != StepKind::Conv || !s1.parts.empty() || !s3.parts.empty()) return false;
            if (s1.kh != 1 || s1.kw != 1 || s1.sh != 1 || s1.sw != 1 || s1.pt || s1.pl || s1.pb || s1.pr || s1.has_in2 || s1.out.c != 128) return false;
            if (s3.kh != 3 || s3.kw != 3 || s3.sh != 1 || s3.sw != 1 || s3.pt != 1 || s3.pl != 1 || s3.pb != 1 || s3.pr != 1 || s3.has_in2 || s3.out.c != 32) return false;
            if (s3.pre_scale_off >= 0 || s1.w_off < 0 || s3.w_off < 0) return false;
            if (!s1.in.f16 || !s1.out.f16 || !s3.out.f16 || s1.in.nchw || s1.out.nchw || s3.out.nchw) return false;
            if (!same_view(s3.in, s1.out) || s3.out.buf != s1.in.buf || s3.out.pitch != s1.in.pitch || s1.out.buf == s1.in.buf) return false;
            if (s3.out.n != s1.in.n || s3.out.h != s1.in.h || s3.out.w != s1.in.w) return false;
            if (s1.in.c < 64 || s1.in.c % 32 || s1.in.pitch % 8 || s1.in.c_off % 8 || s3.out.c_off % 8) return false;
            if (s3.out.c_off < s1.in.c_off + s1.in.c && s3.out.c_off + 32 > s1.in.c_off) return false;
            if ((s1.pre_scale_off >= 0) != (s1.pre_shift_off >= 0)) return false;
            // up to 7 raster tiles: a workgroup per image (chains of layers); larger maps: bands of rows, one layer per launch, while at least one
            // row + halo fits the 8 staged tiles
            // Band mode measured no faster than the two streaming kernels at batch 128 (28x28: 39-58 vs 42-54 us per layer; 56x56: 160-218 vs 93-123 us:
            // the per-band fixed cost -- 72 KB of 3x3 weights into LDS, raster reset, epilogue -- is paid 5 ... 28 times per image), so such steps are
            // only formed on request (IE_DENSE_BAND=1: tests, experiments).
            const int64_t ntiles = (s1.in.h * (s1.in.w + 1) + 31) / 32;
            if (ntiles > 7 && (!env.flag("IE_DENSE_BAND") || (s1.in.w + 1 > 64 && 256 / (s1.in.w + 1) < 3))) return false;
            // T must have no other reader: the fused kernel never writes it to memory
            for (size_t j = i + 2; j < plan.steps.size(); ++j) {
                const Step& q = plan.steps[j];
                if (q.in.buf == s1.out.buf || (q.has_in2 && q.in2.buf == s1.out.buf)) return false;
                if (q.out.buf == s1.out.buf) break;                // the buffer was recycled for another tensor: T was dead by then (liveness pass)
            }
            for (size_t o = 0; o < out_vals.size(); ++o) if (view_of(out_vals[o]).buf == s1.out.buf) return false;
            return true;
        };
        std::vector<Step> blocksteps;
        for (size_t i = 0; i < plan.steps.size();) {
            size_t n = 0;
            while (n < 24 && layer_at(i + 2 * n)) {
                if (n > 0) {
                    if ((plan.steps[i].in.h * (plan.steps[i].in.w + 1) + 31) / 32 > 7) break;       // band mode: one layer per launch
                    const Step& f1 = plan.steps[i], &c1 = plan.steps[i + 2 * n], &p3 = plan.steps[i + 2 * n - 1];
                    if (c1.in.buf != f1.in.buf || c1.in.pitch != f1.in.pitch || c1.in.c_off != f1.in.c_off || c1.in.n != f1.in.n || c1.in.h != f1.in.h || c1.in.w != f1.in.w)
                        break;
                    // a layer's first 192 input channels are requested while the previous layer's 3x3 still runs: they must not be its output
                    if (p3.out.c_off < c1.in.c_off + 192 && p3.out.c_off + 32 > c1.in.c_off) break;
                }
                ++n;
            }
            if (n == 0) { blocksteps.push_back(plan.steps[i]); ++i; continue; }
            Step f = plan.steps[i];
            f.algo = ConvAlgo::DenseBlock;
            f.tile = 1;
            f.splitk = 1;
            f.flops = 0;
            f.bytes = 0;
            f.parts.clear();
            for (size_t q = 0; q < 2 * n; ++q) {
                const Step& ps = plan.steps[i + q];
                f.parts.push_back(ps);
                f.flops += ps.flops;
                f.bytes += ps.bytes;       // SURVEY §8d's per-conv accounting (the bottleneck tensor's write + read is in it although it stays on-chip here)
            }
            f.name = plan.steps[i].name + " ... " + plan.steps[i + 2 * n - 1].name;
            f.out = plan.steps[i + 2 * n - 1].out;             // what the step leaves in memory last (every part's slice is produced by this step)
            blocksteps.push_back(std::move(f));
            i += 2 * n;
        }
        if (blocksteps.size() != plan.steps.size()) {
            std::vector<int> remap(plan.steps.size(), -1);
            for (size_t k = 0; k < blocksteps.size(); ++k) {
                if (blocksteps[k].algo == ConvAlgo::DenseBlock) for (const Step& q : blocksteps[k].parts) remap[size_t(q.idx)] = int(k);
                else remap[size_t(blocksteps[k].idx)] = int(k);
            }
            for (size_t k = 0; k < blocksteps.size(); ++k) {
                Step& st = blocksteps[k];
                st.idx = int(k);
                if (st.in_src >= 0) st.in_src = remap[size_t(st.in_src)];
                if (st.in2_src >= 0) st.in2_src = remap[size_t(st.in2_src)];
            }
            plan.steps = std::move(blocksteps);
        }
    }

    // ---- stem + max pool: the 7x7/s2 stem conv and the 3x3/s2/p1 max pool behind it -> ONE step -------------------------------------
    // Pattern: step i = the stem conv (algo Stem, ReLU'd), step i + 1 = a max pool 3x3 / stride 2 / pad 1 without a prologue that
    // reads exactly step i's output, which nothing else reads.  conv_stem_kernel<POOL> (kernels_stem.hip) pools the conv tile in LDS: the tensor
    // between the two ops (the largest of DenseNet / ResNet) is never written.  parts = {conv, pool}; tile 1 = fused, 0 = the two launches.
    if (!env.get("IE_NO_STEM_POOL")) {
        for (size_t i = 0; i + 1 < plan.steps.size(); ++i) {
            const Step& c = plan.steps[i];
            const Step& pl = plan.steps[i + 1];
            if (c.kind != StepKind::Conv || c.algo != ConvAlgo::Stem || !c.relu || c.has_in2 || !c.parts.empty() || c.out.nchw || c.out.c > 64 || c.out.c % 8) continue;
            if (pl.kind != StepKind::Pool || !pl.pool_max || pl.kh != 3 || pl.kw != 3 || pl.sh != 2 || pl.sw != 2 || pl.pt != 1 || pl.pl != 1 || pl.pb > 1 || pl.pr > 1) continue;
            if (pl.pre_scale_off >= 0 || pl.pre_relu || pl.has_in2 || pl.in_src != int(i)) continue;
            if (pl.in.buf != c.out.buf || pl.in.c_off != c.out.c_off || pl.in.c != c.out.c || pl.in.pitch != c.out.pitch || pl.in.h != c.out.h || pl.in.w != c.out.w) continue;
            if (pl.out.f16 != c.out.f16 || pl.out.f8 != c.out.f8 || pl.out.nchw || pl.out.buf == c.out.buf || pl.out.buf == c.in.buf) continue;
            if (pl.out.h != (c.out.h + 2 - 3) / 2 + 1 || pl.out.w != (c.out.w + 2 - 3) / 2 + 1) continue;
            if (pl.out.pitch % 8 || pl.out.c_off % 8 || (pl.out.f8 && (pl.out.pitch % 16 || pl.out.c_off % 16))) continue;
            // the conv's output must have no other reader: walk the launches behind the pool in execution order (a fused step = its parts) until the
            // buffer is written again -- recycled for another tensor, so the stem's tensor was dead by then (liveness pass)
            bool ok = true, recycled = false;
            auto visit = [&](const Step& u) {
                if (recycled || !ok) return;
                if (u.in.buf == c.out.buf || (u.has_in2 && u.in2.buf == c.out.buf)) ok = false;
                else if (u.out.buf == c.out.buf) recycled = true;
            };
            for (size_t q = i + 2; q < plan.steps.size() && ok && !recycled; ++q) {
                const Step& t = plan.steps[q];
                if (t.parts.empty()) visit(t);
                else for (const Step& tp : t.parts) visit(tp);
            }
            for (size_t o = 0; o < out_vals.size() && ok; ++o) if (view_of(out_vals[o]).buf == c.out.buf) ok = false;
            if (!ok) continue;
            Step f = c;
            f.algo = ConvAlgo::StemPool;
            f.tile = 1;
            f.out = pl.out;
            f.name = c.name + " + " + pl.name;
            f.flops = c.flops + pl.flops;
            f.bytes = double(c.in.numel()) * c.in.esize() + double(pl.out.numel()) * pl.out.esize() + double(c.out.c) * c.kh * c.kw * c.in.c * 4;
            f.parts = {c, pl};
            std::vector<Step> ns;
            ns.reserve(plan.steps.size() - 1);
            for (size_t q = 0; q < plan.steps.size(); ++q) {
                if (q == i + 1) continue;
                ns.push_back(q == i ? f : plan.steps[q]);
            }
            auto remap = [&](int src) { return src < 0 ? src : (size_t(src) > i ? src - 1 : src); };     // i + 1 -> i, everything behind moves up
            for (size_t q = 0; q < ns.size(); ++q) {
                Step& st = ns[q];
                st.idx = int(q);
                st.in_src = remap(st.in_src);
                st.in2_src = remap(st.in2_src);
                for (Step& part : st.parts) { part.in_src = remap(part.in_src); part.in2_src = remap(part.in2_src); }
            }
            // run as two launches, the parts share the fused step's slot: one tensor scale (a max pool keeps its operand's), same producer index
            ns[i].parts[0].idx = int(i);
            ns[i].parts[1].idx = int(i);
            ns[i].parts[1].in_src = int(i);
            plan.steps = std::move(ns);
            break;                                // one stem per graph
        }
    }

    // ---- projection shortcuts (fp8): conv3 + residual where the residual is a 1x1 projection conv -> ONE step of two GEMMs ------------------------
    // Pattern: step j = 1x1/s1 conv C with a fused residual whose producer is step i < j = a plain 1x1 conv P (any stride, no prologue, no
    // residual, no ReLU) read by nothing else.  out = relu(C(a) + P(x)) then runs as two accumulator sets of one launch (kernels_ws8.hip) and
    // P's output -- the largest tensor of the block -- is never written.  P moves down to C's position (everything between them is independent
    // of P's output: its only reader is C).
    if ((precision == Precision::F8 || f8_fusions) && !env.get("IE_NO_DUAL_F8")) {
        for (size_t j = 0; j < plan.steps.size(); ++j) {
            Step& c = plan.steps[j];
            if (c.kind != StepKind::Conv || !c.has_in2 || !c.parts.empty() || c.kh != 1 || c.kw != 1 || c.sh != 1 || c.sw != 1 || c.pt || c.pl || c.pb || c.pr) continue;
            if (c.pre_scale_off >= 0 || c.in.nchw || c.out.nchw || c.in.c % 32 || c.out.c % 32 || c.in2_src < 0 || size_t(c.in2_src) >= j) continue;
            const size_t i = size_t(c.in2_src);
            const Step& pr = plan.steps[i];
            if (pr.kind != StepKind::Conv || !pr.parts.empty() || pr.has_in2 || pr.relu || pr.pre_scale_off >= 0 || pr.kh != 1 || pr.kw != 1 || pr.pt || pr.pl || pr.pb || pr.pr) continue;
            if (pr.in.nchw || pr.in.c % 32 || pr.out.buf != c.in2.buf || pr.out.c_off != c.in2.c_off || pr.out.c != c.in2.c || pr.out.pitch != c.in2.pitch) continue;
            if (pr.out.n != c.out.n || pr.out.h != c.out.h || pr.out.w != c.out.w || pr.out.c != c.out.c) continue;
            if (precision == Precision::F8 && (!pr.in.f8 || !c.in.f8 || !c.out.f8)) continue;
            if (pr.in.buf == c.out.buf || pr.out.buf == c.out.buf) continue;
            // P's output must have no other reader, and P's INPUT must still hold its value at C's position (not recycled in between)
            bool ok = true;
            for (size_t q = i + 1; q < plan.steps.size() && ok; ++q) {
                const Step& t = plan.steps[q];
                if (q != j && (t.in.buf == pr.out.buf || (t.has_in2 && t.in2.buf == pr.out.buf))) ok = false;
                if (q > j && t.out.buf == pr.out.buf) break;
            }
            for (size_t q = i + 1; q < j && ok; ++q)
                if (plan.steps[q].out.buf == pr.in.buf) ok = false;
            for (size_t o = 0; o < out_vals.size() && ok; ++o) if (view_of(out_vals[o]).buf == pr.out.buf) ok = false;
            if (!ok) continue;
            Step f = c;
            f.algo = ConvAlgo::DualF8;
            f.tile = 1;
            f.splitk = 1;
            f.has_in2 = false;
            f.in2 = View();
            f.in2_src = -1;
            f.name = pr.name + " (+) " + c.name;
            f.flops = pr.flops + c.flops;
            f.bytes = pr.bytes + c.bytes;
            f.parts = {pr, c};
            // steps i+1 .. j-1 move up by one, the fused step takes position j - 1 ... simpler: erase i, replace j (indices above i shift by -1)
            std::vector<Step> ns;
            ns.reserve(plan.steps.size() - 1);
            for (size_t q = 0; q < plan.steps.size(); ++q) {
                if (q == i) continue;
                ns.push_back(q == j ? f : plan.steps[q]);
            }
            auto remap = [&](int src) { return src < 0 ? src : (size_t(src) == i ? int(j) - 1 : (size_t(src) > i ? src - 1 : src)); };
            for (size_t q = 0; q < ns.size(); ++q) {
                Step& st = ns[q];
                st.idx = int(q);
                st.in_src = remap(st.in_src);
                st.in2_src = remap(st.in2_src);
                for (Step& part : st.parts) { part.in_src = remap(part.in_src); part.in2_src = remap(part.in2_src); }
            }
            // inside the fused step: the last conv's shortcut comes from the projection part (no plan step of its own any more)
            ns[j - 1].parts[1].in2_src = -1;
            plan.steps = std::move(ns);
            --j;
        }
    }

    // ---- I/O descriptors ---------------------------------------------------------------------------
    for (size_t i = 0; i < m.inputs.size(); ++i) {
        IoDesc d;
        d.name = m.inputs[i].name;
        d.elem_type = m.inputs[i].elem_type;
        d.model_dims = m.inputs[i].dims;
        d.dims = input_shapes[i];
        int v = L.get_val(d.name);
        if (!used[size_t(L.vals[v].root)]) {   // input never consumed: still give it a staging buffer
            plan.buffer_floats.push_back(root_floats(v));
            plan.buffer_f16.push_back(0);
            L.vals[v].buf = int(plan.buffer_floats.size()) - 1;
        }
        d.view = view_of(v);
        plan.inputs.push_back(d);
    }
    for (size_t i = 0; i < m.outputs.size(); ++i) {
        IoDesc d;
        d.name = m.outputs[i].name;
        d.elem_type = m.outputs[i].elem_type;
        d.model_dims = m.outputs[i].dims;
        d.dims = L.vals[out_vals[i]].dims;
        d.view = view_of(out_vals[i]);
        if (!d.view.nchw && (d.view.c_off != 0 || d.view.pitch != d.view.c || d.view.h * d.view.w != 1))
            fail("internal planner error: output " + d.name + " is not dense");
        plan.outputs.push_back(d);
    }
    return plan;
}

static void json_view(std::ostringstream& o, const View& v) {
    o << "{\"buf\":" << v.buf << ",\"n\":" << v.n << ",\"c\":" << v.c << ",\"h\":" << v.h << ",\"w\":" << v.w
      << ",\"c_off\":" << v.c_off << ",\"pitch\":" << v.pitch << ",\"nchw\":" << (v.nchw ? "true" : "false")
      << ",\"f16\":" << (v.f16 ? "true" : "false") << ",\"f8\":" << (v.f8 ? "true" : "false") << "}";
}
static std::string json_escape(const std::string& s) {
    std::string o;
    for (char c : s) { if (c == '"' || c == '\\') o += '\\'; if (uint8_t(c) >= 0x20) o += c; }
    return o;
}

std::string PlanToJson(const Plan& p) {
    static const char* kinds[] = {"conv", "pool", "gap", "eltwise", "copy"};
    static const char* algos[] = {"igemm_vec", "igemm_scalar", "naive", "raster3x3", "ws1x1", "ws3x3", "stem", "direct", "igemm_f8", "dense_fused", "wino3x3", "conv1x1_x6", "dense_block", "dual_f8", "stem_pool"};
    std::ostringstream o;
    o.precision(17);
    o << "{\"inputs\":[";
    for (size_t i = 0; i < p.inputs.size(); ++i) {
        o << (i ? "," : "") << "{\"name\":\"" << json_escape(p.inputs[i].name) << "\",\"dims\":[";
        for (size_t k = 0; k < p.inputs[i].dims.size(); ++k) o << (k ? "," : "") << p.inputs[i].dims[k];
        o << "],\"view\":"; json_view(o, p.inputs[i].view); o << "}";
    }
    o << "],\"outputs\":[";
    for (size_t i = 0; i < p.outputs.size(); ++i) {
        o << (i ? "," : "") << "{\"name\":\"" << json_escape(p.outputs[i].name) << "\",\"dims\":[";
        for (size_t k = 0; k < p.outputs[i].dims.size(); ++k) o << (k ? "," : "") << p.outputs[i].dims[k];
        o << "],\"view\":"; json_view(o, p.outputs[i].view); o << "}";
    }
    o << "],\"buffers\":[";
    for (size_t i = 0; i < p.buffer_floats.size(); ++i) o << (i ? "," : "") << p.buffer_floats[i];
    o << "],\"precision\":\"" << (p.precision == Precision::F16 ? "fp16" : (p.precision == Precision::F8 ? "fp8" : "fp32")) << "\",\"activation_bytes\":" << p.activation_bytes()
      << ",\"workspace_floats\":" << p.workspace_floats << ",\"weight_floats\":" << p.weights.size() << ",\"total_flops\":" << p.total_flops
      << ",\"total_bytes\":" << p.total_bytes << ",\"steps\":[";
    for (size_t i = 0; i < p.steps.size(); ++i) {
        const Step& s = p.steps[i];
        o << (i ? "," : "") << "{\"kind\":\"" << kinds[int(s.kind)] << "\",\"name\":\"" << json_escape(s.name) << "\",\"in\":";
        json_view(o, s.in);
        if (s.has_in2) { o << ",\"in2\":"; json_view(o, s.in2); }
        if (s.kind == StepKind::Conv) o << ",\"residual\":" << (s.has_in2 ? "true" : "false");
        o << ",\"out\":"; json_view(o, s.out);
        o << ",\"k\":[" << s.kh << "," << s.kw << "],\"stride\":[" << s.sh << "," << s.sw << "],\"pads\":[" << s.pt << ","
          << s.pl << "," << s.pb << "," << s.pr << "]";
        o << ",\"pre\":" << (s.pre_scale_off >= 0 ? "true" : "false") << ",\"pre_relu\":" << (s.pre_relu ? "true" : "false")
          << ",\"relu\":" << (s.relu ? "true" : "false") << ",\"bias\":" << (s.bias_off >= 0 ? "true" : "false");
        if (s.kind == StepKind::Conv) o << ",\"algo\":\"" << algos[int(s.algo)] << "\",\"tile\":" << s.tile << ",\"splitk\":" << s.splitk;
        if (s.kind == StepKind::Pool) o << ",\"max\":" << (s.pool_max ? "true" : "false");
        if (!s.parts.empty()) {
            Plan sub;
            sub.steps = s.parts;
            const std::string js = PlanToJson(sub);
            const size_t b = js.find("\"steps\":[");
            o << ",\"parts\":" << js.substr(b + 8, js.size() - (b + 8) - 1);
        }
        o << ",\"idx\":" << s.idx << ",\"in_src\":" << s.in_src << ",\"in2_src\":" << s.in2_src << ",\"w_off\":" << s.w_off << ",\"bias_off\":" << s.bias_off
          << ",\"pre_scale_off\":" << s.pre_scale_off << ",\"pre_shift_off\":" << s.pre_shift_off << ",\"count_include_pad\":" << (s.count_include_pad ? "true" : "false");
        o << ",\"flops\":" << s.flops << ",\"bytes\":" << s.bytes << "}";
    }
    o << "]}";
    return o.str();
}

}  // namespace ie
