#include "onnx_reader.h"

#include <cstring>
#include <fstream>
#include <stdexcept>

namespace ie {
namespace {

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        int shift = 0;
        while (true) {
            if (p >= end || shift > 63) throw std::runtime_error("ONNX parse error: truncated varint");
            uint8_t c = *p++;
            v |= uint64_t(c & 0x7F) << shift;
            if (!(c & 0x80)) return v;
            shift += 7;
        }
    }
    Reader sub(uint64_t n) {
        if (uint64_t(end - p) < n) throw std::runtime_error("ONNX parse error: truncated field");
        Reader r{p, p + n};
        p += n;
        return r;
    }
    void skip(int wire) {
        switch (wire) {
            case 0: varint(); break;
            case 1: sub(8); break;
            case 2: sub(varint()); break;
            case 5: sub(4); break;
            default: throw std::runtime_error("ONNX parse error: unsupported wire type");
        }
    }
    std::string str(Reader r) { return std::string(reinterpret_cast<const char*>(r.p), r.end - r.p); }
};

float half_to_float(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1F, m = h & 0x3FF, o;
    if (e == 0) {
        if (m == 0) o = s << 31;
        else {
            e = 127 - 15 + 1;
            while (!(m & 0x400)) { m <<= 1; --e; }
            o = (s << 31) | (e << 23) | ((m & 0x3FF) << 13);
        }
    } else if (e == 31) o = (s << 31) | 0x7F800000u | (m << 13);
    else o = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &o, 4);
    return f;
}

// repeated scalar: accepts packed (wire 2) and unpacked (wire 0) encodings.
void read_varints(Reader& r, int wire, std::vector<int64_t>& out) {
    if (wire == 0) { out.push_back(int64_t(r.varint())); return; }
    if (wire != 2) throw std::runtime_error("ONNX parse error: bad repeated-int encoding");
    Reader s = r.sub(r.varint());
    while (!s.done()) out.push_back(int64_t(s.varint()));
}
void read_floats(Reader& r, int wire, std::vector<float>& out) {
    if (wire == 5) { Reader s = r.sub(4); float f; std::memcpy(&f, s.p, 4); out.push_back(f); return; }
    if (wire != 2) throw std::runtime_error("ONNX parse error: bad repeated-float encoding");
    Reader s = r.sub(r.varint());
    size_t n = (s.end - s.p) / 4;
    size_t o = out.size();
    out.resize(o + n);
    std::memcpy(out.data() + o, s.p, n * 4);
}

OnnxTensor parse_tensor(Reader r) {
    OnnxTensor t;
    std::vector<int64_t> i32, i64;
    std::vector<float> f32;
    std::vector<double> f64;
    Reader raw{nullptr, nullptr};
    bool has_raw = false;
    while (!r.done()) {
        uint64_t key = r.varint();
        int field = int(key >> 3), wire = int(key & 7);
        switch (field) {
            case 1: read_varints(r, wire, t.dims); break;
            case 2: t.dtype = int(r.varint()); break;
            case 4: read_floats(r, wire, f32); break;
            case 5: read_varints(r, wire, i32); break;
            case 7: read_varints(r, wire, i64); break;
            case 8: t.name = r.str(r.sub(r.varint())); break;
            case 9: raw = r.sub(r.varint()); has_raw = true; break;
            case 10: {
                if (wire == 1) { Reader s = r.sub(8); double d; std::memcpy(&d, s.p, 8); f64.push_back(d); }
                else { Reader s = r.sub(r.varint()); size_t n = (s.end - s.p) / 8; size_t o = f64.size();
                       f64.resize(o + n); std::memcpy(f64.data() + o, s.p, n * 8); }
                break;
            }
            case 13: case 14: {
                if (field == 14 && wire == 0) { if (r.varint() == 1) throw std::runtime_error("ONNX external data is not supported"); }
                else r.skip(wire);
                break;
            }
            default: r.skip(wire);
        }
    }
    int64_t n = t.numel();
    if (n < 0) throw std::runtime_error("ONNX parse error: negative tensor dims in initializer " + t.name);
    auto need = [&](size_t have, size_t elem) {
        if (have != size_t(n) * elem) throw std::runtime_error("ONNX parse error: initializer " + t.name + " payload size mismatch");
    };
    switch (t.dtype) {
        case ONNX_FLOAT:
            if (has_raw) { need(raw.end - raw.p, 4); t.f.resize(n); std::memcpy(t.f.data(), raw.p, n * 4); }
            else { need(f32.size(), 1); t.f = std::move(f32); }
            break;
        case ONNX_DOUBLE:
            if (has_raw) { need(raw.end - raw.p, 8); f64.resize(n); std::memcpy(f64.data(), raw.p, n * 8); }
            else need(f64.size(), 1);
            t.f.assign(f64.begin(), f64.end());
            break;
        case ONNX_FLOAT16:
            t.f.resize(n);
            if (has_raw) { need(raw.end - raw.p, 2); for (int64_t k = 0; k < n; ++k) { uint16_t h; std::memcpy(&h, raw.p + 2 * k, 2); t.f[k] = half_to_float(h); } }
            else { need(i32.size(), 1); for (int64_t k = 0; k < n; ++k) t.f[k] = half_to_float(uint16_t(i32[k])); }
            break;
        case ONNX_INT64:
            if (has_raw) { need(raw.end - raw.p, 8); t.i.resize(n); std::memcpy(t.i.data(), raw.p, n * 8); }
            else { need(i64.size(), 1); t.i = std::move(i64); }
            break;
        case ONNX_INT32:
            if (has_raw) { need(raw.end - raw.p, 4); t.i.resize(n); for (int64_t k = 0; k < n; ++k) { int32_t v; std::memcpy(&v, raw.p + 4 * k, 4); t.i[k] = v; } }
            else { need(i32.size(), 1); t.i.resize(n); for (int64_t k = 0; k < n; ++k) t.i[k] = int32_t(i32[k]); }
            break;
        default:
            throw std::runtime_error("ONNX initializer " + t.name + ": unsupported data type " + std::to_string(t.dtype));
    }
    return t;
}

OnnxAttr parse_attr(Reader r) {
    OnnxAttr a;
    while (!r.done()) {
        uint64_t key = r.varint();
        int field = int(key >> 3), wire = int(key & 7);
        switch (field) {
            case 1: a.name = r.str(r.sub(r.varint())); break;
            case 2: { Reader s = r.sub(4); std::memcpy(&a.f, s.p, 4); break; }
            case 3: a.i = int64_t(r.varint()); break;
            case 4: a.s = r.str(r.sub(r.varint())); break;
            case 5: a.t = parse_tensor(r.sub(r.varint())); a.has_t = true; break;
            case 7: read_floats(r, wire, a.floats); break;
            case 8: read_varints(r, wire, a.ints); break;
            case 20: a.type = int(r.varint()); break;
            default: r.skip(wire);
        }
    }
    return a;
}

OnnxNode parse_node(Reader r) {
    OnnxNode n;
    while (!r.done()) {
        uint64_t key = r.varint();
        int field = int(key >> 3), wire = int(key & 7);
        switch (field) {
            case 1: n.inputs.push_back(r.str(r.sub(r.varint()))); break;
            case 2: n.outputs.push_back(r.str(r.sub(r.varint()))); break;
            case 3: n.name = r.str(r.sub(r.varint())); break;
            case 4: n.op = r.str(r.sub(r.varint())); break;
            case 5: { OnnxAttr a = parse_attr(r.sub(r.varint())); n.attrs[a.name] = std::move(a); break; }
            default: r.skip(wire);
        }
    }
    return n;
}

OnnxValueInfo parse_value_info(Reader r) {
    OnnxValueInfo v;
    while (!r.done()) {
        uint64_t key = r.varint();
        int field = int(key >> 3), wire = int(key & 7);
        if (field == 1) v.name = r.str(r.sub(r.varint()));
        else if (field == 2) {                       // TypeProto
            Reader tp = r.sub(r.varint());
            while (!tp.done()) {
                uint64_t k2 = tp.varint();
                if ((k2 >> 3) == 1 && (k2 & 7) == 2) {   // tensor_type
                    Reader tt = tp.sub(tp.varint());
                    while (!tt.done()) {
                        uint64_t k3 = tt.varint();
                        if ((k3 >> 3) == 1) v.elem_type = int(tt.varint());
                        else if ((k3 >> 3) == 2) {       // TensorShapeProto
                            Reader sh = tt.sub(tt.varint());
                            while (!sh.done()) {
                                uint64_t k4 = sh.varint();
                                if ((k4 >> 3) == 1 && (k4 & 7) == 2) {
                                    Reader dm = sh.sub(sh.varint());
                                    int64_t d = -1;
                                    while (!dm.done()) {
                                        uint64_t k5 = dm.varint();
                                        if ((k5 >> 3) == 1) d = int64_t(dm.varint());
                                        else dm.skip(int(k5 & 7));
                                    }
                                    v.dims.push_back(d);
                                } else sh.skip(int(k4 & 7));
                            }
                        } else tt.skip(int(k3 & 7));
                    }
                } else tp.skip(int(k2 & 7));
            }
        } else r.skip(wire);
    }
    return v;
}

}  // namespace

OnnxModel ParseOnnx(const uint8_t* data, size_t size) {
    OnnxModel m;
    Reader r{data, data + size};
    bool have_graph = false;
    std::vector<OnnxValueInfo> all_inputs;
    while (!r.done()) {
        uint64_t key = r.varint();
        int field = int(key >> 3), wire = int(key & 7);
        if (field == 1 && wire == 0) m.ir_version = int64_t(r.varint());
        else if (field == 2 && wire == 2) m.producer = r.str(r.sub(r.varint()));
        else if (field == 8 && wire == 2) {
            Reader o = r.sub(r.varint());
            std::string domain;
            int64_t ver = 0;
            while (!o.done()) {
                uint64_t k2 = o.varint();
                if ((k2 >> 3) == 1) domain = o.str(o.sub(o.varint()));
                else if ((k2 >> 3) == 2) ver = int64_t(o.varint());
                else o.skip(int(k2 & 7));
            }
            if (domain.empty() || domain == "ai.onnx") m.opset = ver;
        } else if (field == 7 && wire == 2) {
            have_graph = true;
            Reader g = r.sub(r.varint());
            while (!g.done()) {
                uint64_t k2 = g.varint();
                int f2 = int(k2 >> 3), w2 = int(k2 & 7);
                if (f2 == 1) m.nodes.push_back(parse_node(g.sub(g.varint())));
                else if (f2 == 2) m.graph_name = g.str(g.sub(g.varint()));
                else if (f2 == 5) { OnnxTensor t = parse_tensor(g.sub(g.varint())); std::string nm = t.name; m.initializers[nm] = std::move(t); }
                else if (f2 == 11) all_inputs.push_back(parse_value_info(g.sub(g.varint())));
                else if (f2 == 12) m.outputs.push_back(parse_value_info(g.sub(g.varint())));
                else g.skip(w2);
            }
        } else r.skip(wire);
    }
    if (!have_graph) throw std::runtime_error("ONNX parse error: no graph in model");
    for (auto& vi : all_inputs)
        if (!m.initializers.count(vi.name)) m.inputs.push_back(vi);
    return m;
}

OnnxModel LoadOnnxFile(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open ONNX file: " + path);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return ParseOnnx(buf.data(), buf.size());
}

}  // namespace ie
