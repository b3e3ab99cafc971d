// gob.cpp -- reader for indexes written by the reference's `groot index` (Go encoding/gob streams)
//
//   <indexDir>/groot.gg    gob of pipeline.Info            (src/pipeline/runtime.go:15-28,64-72; cmd/index.go:131)
//   <indexDir>/groot.lshe  gob of lshe.ContainmentIndex    (src/lshe/lshe.go:38-44,71-92;       cmd/index.go:130)
//
// The decoder is generic (it follows the type definitions carried in the stream, as the gob wire format
// prescribes) and is then mapped by FIELD NAME onto the flat index of include/groot_index.h.  Wire format, as
// published in the Go documentation of encoding/gob:
//   * unsigned: one byte if < 128, else a byte holding the negated byte count followed by the big-endian bytes;
//   * signed:   unsigned u with the sign in bit 0: (u & 1) ? ~(u >> 1) : (u >> 1);
//   * float64:  IEEE bits byte-reversed, sent as unsigned;  bool: unsigned 0/1;  string / []byte: length + bytes;
//   * struct:   (field-number delta, value)* terminated by delta 0, zero-valued fields omitted;
//   * slice / array: count + elements;  map: count + (key, element)*;  pointers are flattened;
//   * stream:   messages = length + (negative type id + wireType definition | type id + value), user ids from 65.
// Go is not available where this was written: the decoder is pinned on the byte vectors printed in the gob
// documentation (tests/test_gob.py) and on round trips through an independent encoder, NOT on a Go-written file.
#include <algorithm>
#include <cinttypes>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <stdexcept>
#include <unordered_map>

#include "host_common.hpp"

namespace groot {
namespace gob {

enum Kind { K_BOOL, K_INT, K_UINT, K_FLOAT, K_BYTES, K_STRING, K_STRUCT, K_SLICE, K_ARRAY, K_MAP, K_OTHER };

struct Type {
    Kind kind = K_OTHER;
    std::string name;
    int elem = 0, key = 0;
    int64_t len = 0;
    std::vector<std::pair<std::string, int>> fields;
};

static bool numeric(Kind k) { return k == K_BOOL || k == K_INT || k == K_UINT || k == K_FLOAT; }

struct Value {
    Kind kind = K_OTHER;
    uint64_t u = 0;                 // bool / uint; int as two's complement; float as IEEE bits
    std::string s;                  // string / bytes
    std::vector<uint64_t> nums;     // slice/array of numeric elements; map numeric->numeric as k,v,k,v
    std::vector<Value> items;       // struct: one per PRESENT field (see field_idx); slice of non-numeric; map as k,v,k,v
    std::vector<int> field_idx;     // struct: index into Type::fields of items[i]
    const Type *type = nullptr;
    bool packed = false;            // nums in use rather than items

    const Value *field(const char *name) const
    {
        for (size_t i = 0; i < items.size(); i++)
            if (type->fields[field_idx[i]].first == name) return &items[i];
        return nullptr;
    }
    uint64_t uint_field(const char *name, uint64_t dflt = 0) const
    {
        const Value *v = field(name);
        return v ? v->u : dflt;
    }
    double as_double() const { double d; memcpy(&d, &u, 8); return d; }
    size_t count() const
    {
        const size_t n = packed ? nums.size() : items.size();
        return kind == K_MAP ? n / 2 : n;
    }
};

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// called for every entry of a map-typed field of the top-level struct; return true to consume it (not stored)
using MapHook = std::function<bool(const std::string &field, Value &key, Value &val)>;

class Decoder {
  public:
    Decoder(const uint8_t *p, size_t n) : p_(p), end_(p + n) { bootstrap(); }

    bool at_end() const { return p_ >= end_; }

    const Type &type_of(int id) const
    {
        auto it = types_.find(id);
        if (it == types_.end()) throw Error("gob: value of undefined type id " + std::to_string(id));
        return it->second;
    }

    // decodes messages until the next top-level value (type definitions are absorbed on the way)
    Value next(const MapHook *hook = nullptr)
    {
        for (;;) {
            if (p_ >= end_) throw Error("gob: stream ended before a value was found");
            const uint64_t len = get_uint();
            if (len == 0 || len > (uint64_t)(end_ - p_)) throw Error("gob: message length exceeds the file");
            const uint8_t *save_end = end_;
            const uint8_t *msg_end = p_ + len;
            end_ = msg_end;
            const int64_t id = get_int();
            if (id < 0) {
                define_type((int)-id);
                if (p_ != msg_end) throw Error("gob: trailing bytes after a type definition");
                end_ = save_end;
                continue;
            }
            const Type &t = type_of((int)id);
            Value v;
            if (t.kind == K_STRUCT) {
                decode_struct(t, v, hook, 0);
            } else {
                if (get_uint() != 0) throw Error("gob: non-struct top-level value without its zero delta");
                decode(t, v, nullptr, 1);
            }
            if (p_ != msg_end) throw Error("gob: trailing bytes after a value");
            end_ = save_end;
            return v;
        }
    }

  private:
    const uint8_t *p_, *end_;
    std::map<int, Type> types_;

    void need(size_t n) const
    {
        if ((size_t)(end_ - p_) < n) throw Error("gob: truncated");
    }
    uint64_t get_uint()
    {
        need(1);
        const uint8_t b = *p_++;
        if (b < 0x80) return b;
        const unsigned n = 256u - b;   // negated byte count
        if (n == 0 || n > 8) throw Error("gob: bad unsigned integer length");
        need(n);
        uint64_t v = 0;
        for (unsigned i = 0; i < n; i++) v = (v << 8) | *p_++;
        return v;
    }
    int64_t get_int()
    {
        const uint64_t u = get_uint();
        return (u & 1) ? (int64_t)~(u >> 1) : (int64_t)(u >> 1);
    }
    uint64_t get_float_bits()
    {
        uint64_t u = get_uint(), r = 0;
        for (int i = 0; i < 8; i++) { r = (r << 8) | (u & 0xFF); u >>= 8; }
        return r;
    }
    uint64_t get_count(size_t min_bytes_each)
    {
        const uint64_t n = get_uint();
        if (min_bytes_each && n > (uint64_t)(end_ - p_) / min_bytes_each) throw Error("gob: element count exceeds the message");
        return n;
    }

    void bootstrap()
    {
        auto basic = [&](int id, Kind k, const char *name) { Type t; t.kind = k; t.name = name; types_[id] = t; };
        basic(1, K_BOOL, "bool"); basic(2, K_INT, "int"); basic(3, K_UINT, "uint"); basic(4, K_FLOAT, "float");
        basic(5, K_BYTES, "bytes"); basic(6, K_STRING, "string"); basic(7, K_OTHER, "complex"); basic(8, K_OTHER, "interface");
        auto strct = [&](int id, const char *name, std::vector<std::pair<std::string, int>> f) {
            Type t; t.kind = K_STRUCT; t.name = name; t.fields = std::move(f); types_[id] = t;
        };
        // the self-describing part of the format: wireType and its components (ids fixed by the gob package)
        strct(16, "wireType", {{"ArrayT", 17}, {"SliceT", 19}, {"StructT", 20}, {"MapT", 23}});
        strct(17, "arrayType", {{"CommonType", 18}, {"Elem", 2}, {"Len", 2}});
        strct(18, "CommonType", {{"Name", 6}, {"Id", 2}});
        strct(19, "sliceType", {{"CommonType", 18}, {"Elem", 2}});
        strct(20, "structType", {{"CommonType", 18}, {"Field", 22}});
        strct(21, "fieldType", {{"Name", 6}, {"Id", 2}});
        Type fs; fs.kind = K_SLICE; fs.name = "[]*gob.fieldType"; fs.elem = 21; types_[22] = fs;
        strct(23, "mapType", {{"CommonType", 18}, {"Key", 2}, {"Elem", 2}});
    }

    void define_type(int id)
    {
        Value w;
        decode_struct(type_of(16), w, nullptr, 1);
        if (w.items.size() != 1) throw Error("gob: type definition with an unsupported wireType (GobEncoder/Marshaler types are not handled)");
        const Value &d = w.items[0];
        const std::string which = w.type->fields[w.field_idx[0]].first;
        Type t;
        if (const Value *c = d.field("CommonType"))
            if (const Value *n = c->field("Name")) t.name = n->s;
        auto id_of = [&](const char *f) { const Value *v = d.field(f); return v ? (int)(int64_t)v->u : 0; };
        if (which == "StructT") {
            t.kind = K_STRUCT;
            if (const Value *fl = d.field("Field"))
                for (const Value &f : fl->items) {
                    const Value *n = f.field("Name");
                    const Value *i = f.field("Id");
                    t.fields.emplace_back(n ? n->s : std::string(), i ? (int)(int64_t)i->u : 0);
                }
        } else if (which == "SliceT") {
            t.kind = K_SLICE; t.elem = id_of("Elem");
        } else if (which == "ArrayT") {
            t.kind = K_ARRAY; t.elem = id_of("Elem");
            const Value *l = d.field("Len");
            t.len = l ? (int64_t)l->u : 0;
        } else if (which == "MapT") {
            t.kind = K_MAP; t.key = id_of("Key"); t.elem = id_of("Elem");
        } else {
            throw Error("gob: unsupported wire type " + which);
        }
        types_[id] = t;
    }

    void decode_scalar(Kind k, uint64_t &u)
    {
        switch (k) {
        case K_BOOL: case K_UINT: u = get_uint(); break;
        case K_INT: u = (uint64_t)get_int(); break;
        case K_FLOAT: u = get_float_bits(); break;
        default: throw Error("gob: not a scalar");
        }
    }

    void decode_struct(const Type &t, Value &out, const MapHook *hook, int depth)
    {
        out.kind = K_STRUCT;
        out.type = &t;
        int64_t fieldnum = -1;
        for (;;) {
            const uint64_t delta = get_uint();
            if (delta == 0) break;
            fieldnum += (int64_t)delta;
            if (fieldnum >= (int64_t)t.fields.size()) throw Error("gob: field number out of range in " + t.name);
            const Type &ft = type_of(t.fields[(size_t)fieldnum].second);
            Value v;
            if (hook && depth == 0 && ft.kind == K_MAP) decode_map(ft, v, hook, &t.fields[(size_t)fieldnum].first);
            else decode(ft, v, nullptr, depth + 1);
            out.items.push_back(std::move(v));
            out.field_idx.push_back((int)fieldnum);
        }
    }

    void decode_map(const Type &t, Value &out, const MapHook *hook, const std::string *field)
    {
        out.kind = K_MAP;
        out.type = &t;
        const Type &kt = type_of(t.key), &et = type_of(t.elem);
        const uint64_t n = get_count(2);
        if (!hook && numeric(kt.kind) && numeric(et.kind)) {
            out.packed = true;
            out.nums.resize(n * 2);
            for (uint64_t i = 0; i < n; i++) {
                decode_scalar(kt.kind, out.nums[2 * i]);
                decode_scalar(et.kind, out.nums[2 * i + 1]);
            }
            return;
        }
        for (uint64_t i = 0; i < n; i++) {
            Value k, v;
            decode(kt, k, nullptr, 2);
            decode(et, v, nullptr, 2);
            if (hook && (*hook)(*field, k, v)) continue;
            out.items.push_back(std::move(k));
            out.items.push_back(std::move(v));
        }
    }

    void decode(const Type &t, Value &out, const MapHook *, int depth)
    {
        if (depth > 64) throw Error("gob: nesting too deep");
        out.kind = t.kind;
        out.type = &t;
        switch (t.kind) {
        case K_BOOL: case K_INT: case K_UINT: case K_FLOAT: decode_scalar(t.kind, out.u); break;
        case K_BYTES: case K_STRING: {
            const uint64_t n = get_count(1);
            out.s.assign((const char *)p_, (size_t)n);
            p_ += n;
            break;
        }
        case K_STRUCT: decode_struct(t, out, nullptr, depth); break;
        case K_MAP: decode_map(t, out, nullptr, nullptr); break;
        case K_SLICE: case K_ARRAY: {
            const Type &et = type_of(t.elem);
            const uint64_t n = get_count(1);
            if (t.kind == K_ARRAY && (int64_t)n != t.len) throw Error("gob: array length mismatch");
            if (numeric(et.kind)) {
                out.packed = true;
                out.nums.resize(n);
                for (uint64_t i = 0; i < n; i++) decode_scalar(et.kind, out.nums[i]);
            } else {
                out.items.resize(n);
                for (uint64_t i = 0; i < n; i++) decode(et, out.items[i], nullptr, depth + 1);
            }
            break;
        }
        default: throw Error("gob: unsupported type " + t.name + " (interface, complex and GobEncoder values are not handled)");
        }
    }
};

// ---- JSON rendering of a decoded value (debugging + the test pins) ---------------------------------
static void json_string(const std::string &s, std::string &o)
{
    o += '"';
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c < 0x20 || c >= 0x7F) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += (char)c;
    }
    o += '"';
}
static void json_scalar(Kind k, uint64_t u, std::string &o)
{
    char b[40];
    if (k == K_BOOL) { o += u ? "true" : "false"; return; }
    if (k == K_INT) snprintf(b, sizeof b, "%" PRId64, (int64_t)u);
    else if (k == K_UINT) snprintf(b, sizeof b, "%" PRIu64, u);
    else { double d; memcpy(&d, &u, 8); snprintf(b, sizeof b, "%.17g", d); }
    o += b;
}
static void to_json(const Value &v, const Decoder &dec, std::string &o)
{
    switch (v.kind) {
    case K_BOOL: case K_INT: case K_UINT: case K_FLOAT: json_scalar(v.kind, v.u, o); break;
    case K_BYTES: case K_STRING: json_string(v.s, o); break;
    case K_STRUCT:
        o += '{';
        for (size_t i = 0; i < v.items.size(); i++) {
            if (i) o += ',';
            json_string(v.type->fields[v.field_idx[i]].first, o);
            o += ':';
            to_json(v.items[i], dec, o);
        }
        o += '}';
        break;
    case K_SLICE: case K_ARRAY:
        o += '[';
        if (v.packed) {
            const Kind ek = dec.type_of(v.type->elem).kind;
            for (size_t i = 0; i < v.nums.size(); i++) { if (i) o += ','; json_scalar(ek, v.nums[i], o); }
        } else {
            for (size_t i = 0; i < v.items.size(); i++) { if (i) o += ','; to_json(v.items[i], dec, o); }
        }
        o += ']';
        break;
    case K_MAP:
        o += '[';
        if (v.packed) {
            const Kind kk = dec.type_of(v.type->key).kind, ek = dec.type_of(v.type->elem).kind;
            for (size_t i = 0; i + 1 < v.nums.size(); i += 2) {
                if (i) o += ',';
                o += '['; json_scalar(kk, v.nums[i], o); o += ','; json_scalar(ek, v.nums[i + 1], o); o += ']';
            }
        } else {
            for (size_t i = 0; i + 1 < v.items.size(); i += 2) {
                if (i) o += ',';
                o += '['; to_json(v.items[i], dec, o); o += ','; to_json(v.items[i + 1], dec, o); o += ']';
            }
        }
        o += ']';
        break;
    default: o += "null";
    }
}

} // namespace gob
} // namespace groot

// ---- writer: the same wire format, for index directories the reference can load ---------------------
namespace groot {
namespace gob {

struct GType {
    enum { BASIC, STRUCT, SLICE, MAP } kind = BASIC;
    int basic = 0;                                                  // 1 bool 2 int 3 uint 4 float 5 bytes 6 string
    std::string name;
    std::vector<std::pair<std::string, const GType *>> fields;     // STRUCT
    const GType *elem = nullptr, *key = nullptr;                   // SLICE / MAP
};

static void put_uint(std::string &o, uint64_t v)
{
    if (v < 128) { o.push_back((char)v); return; }
    char b[8];
    int n = 0;
    for (uint64_t x = v; x; x >>= 8) n++;
    for (int i = 0; i < n; i++) b[i] = (char)(v >> (8 * (n - 1 - i)));
    o.push_back((char)(256 - n));
    o.append(b, (size_t)n);
}
static void put_int(std::string &o, int64_t i) { put_uint(o, i < 0 ? ((~(uint64_t)i) << 1) | 1u : (uint64_t)i << 1); }
static void put_float(std::string &o, double d)
{
    uint64_t u, r = 0;
    memcpy(&u, &d, 8);
    for (int i = 0; i < 8; i++) { r = (r << 8) | (u & 0xFF); u >>= 8; }
    put_uint(o, r);
}
static void put_bytes(std::string &o, const void *p, size_t n) { put_uint(o, n); o.append((const char *)p, n); }

// one gob stream: type definitions are sent the first time a type is used (definition first, then the types it refers to)
class Encoder {
  public:
    std::string out;
    int id_of(const GType *t)
    {
        if (t->kind == GType::BASIC) return t->basic;
        auto it = ids_.find(t);
        if (it != ids_.end()) return it->second;
        const int id = next_id_++;
        ids_[t] = id;
        send_type(t);
        return id;
    }
    // a top-level struct value: `body` holds its fields (as StructWriter leaves them, terminator included)
    void message(const GType *t, const std::string &body)
    {
        std::string head;
        put_int(head, id_of(t));
        put_uint(out, head.size() + body.size());
        out += head;
        out += body;
    }

  private:
    std::map<const GType *, int> ids_;
    int next_id_ = 65;
    int ref(const GType *t) const { return t->kind == GType::BASIC ? t->basic : ids_.at(t); }
    void common(std::string &w, const std::string &name, int id)
    {
        if (!name.empty()) { w.push_back(1); put_bytes(w, name.data(), name.size()); w.push_back(1); }
        else w.push_back(2);
        put_int(w, id);
        w.push_back(0);
    }
    void send_type(const GType *t)
    {
        const int id = ids_.at(t);
        std::vector<const GType *> parts, inner;
        if (t->kind == GType::STRUCT) for (auto &f : t->fields) parts.push_back(f.second);
        else if (t->kind == GType::SLICE) parts = {t->elem};
        else parts = {t->key, t->elem};
        for (const GType *p : parts)
            if (p->kind != GType::BASIC && !ids_.count(p)) { ids_[p] = next_id_++; inner.push_back(p); }
        std::string w;
        if (t->kind == GType::STRUCT) {
            w.push_back(3); w.push_back(1);
            common(w, t->name, id);
            if (!t->fields.empty()) {
                w.push_back(1);
                put_uint(w, t->fields.size());
                for (auto &f : t->fields) {
                    w.push_back(1); put_bytes(w, f.first.data(), f.first.size());
                    w.push_back(1); put_int(w, ref(f.second));
                    w.push_back(0);
                }
            }
            w.push_back(0);
        } else {
            w.push_back(t->kind == GType::SLICE ? 2 : 4); w.push_back(1);
            common(w, t->name, id);
            for (const GType *p : parts) { w.push_back(1); put_int(w, ref(p)); }
            w.push_back(0);
        }
        w.push_back(0);
        std::string msg;
        put_int(msg, -id);
        msg += w;
        put_uint(out, msg.size());
        out += msg;
        for (const GType *p : inner) send_type(p);
    }
};

// fields of one struct value: call field(i) before writing the value of a non-zero field i (ascending), end() last
struct StructWriter {
    std::string &b;
    int last = -1;
    explicit StructWriter(std::string &buf) : b(buf) {}
    void field(int i) { put_uint(b, (uint64_t)(i - last)); last = i; }
    void u(int i, uint64_t v) { if (v) { field(i); put_uint(b, v); } }
    void i64(int i, int64_t v) { if (v) { field(i); put_int(b, v); } }
    void f(int i, double v) { if (v != 0.0) { field(i); put_float(b, v); } }
    void str(int i, const void *p, size_t n) { if (n) { field(i); put_bytes(b, p, n); } }
    void end() { b.push_back(0); }
};

} // namespace gob
} // namespace groot

using namespace groot;
using gob::Value;

namespace {

struct Blob {
    std::vector<uint8_t> bytes;
    int load(const char *path)
    {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (!f) return set_error(GROOT_E_IO, "cannot open %s", path);
        const std::streamoff n = f.tellg();
        if (n <= 0) return set_error(GROOT_E_FORMAT, "%s appears empty", path);   // runtime.go:84-86, lshe.go:102-104
        bytes.resize((size_t)n);
        f.seekg(0);
        if (!f.read((char *)bytes.data(), n)) return set_error(GROOT_E_IO, "cannot read %s", path);
        return GROOT_OK;
    }
};

uint32_t checked_u32(uint64_t v, const char *what)
{
    if (v > 0xFFFFFFFFull) throw gob::Error(std::string("gob index: ") + what + " does not fit 32 bits");
    return (uint32_t)v;
}

// WindowLookup keys are "g<graph>n<node>o<offset>-<i>" (src/pipeline/index.go:199, graph.go:352): <i> orders the
// windows that share a start position, which is the last component of the canonical window order
bool parse_lookup_index(const std::string &key, uint64_t &i)
{
    const size_t dash = key.rfind('-');
    if (dash == std::string::npos || dash + 1 >= key.size()) return false;
    i = 0;
    for (size_t p = dash + 1; p < key.size(); p++) {
        if (key[p] < '0' || key[p] > '9') return false;
        i = i * 10 + (uint64_t)(key[p] - '0');
    }
    return true;
}

struct LoadedWindow {
    Window w;
    uint64_t list_index;
};

void graph_from_value(const Value &gv, uint32_t key, Graph &g)
{
    if (gv.kind != gob::K_STRUCT) throw gob::Error("gob index: Store element is not a struct");
    g.id = (uint32_t)gv.uint_field("GraphID");
    if (g.id != key) throw gob::Error("gob index: Store key differs from GraphID");
    g.masked = gv.uint_field("Masked") != 0;
    // Paths map[uint32][]byte, Lengths map[uint32]int (graph.go:22-23): path ids must be 0..n-1
    const Value *paths = gv.field("Paths");
    const Value *lengths = gv.field("Lengths");
    const size_t np = paths ? paths->count() : 0;
    g.path_names.assign(np, std::string());
    g.path_len.assign(np, 0);
    std::vector<uint8_t> seen(np, 0);
    for (size_t i = 0; i < np; i++) {
        const uint64_t id = paths->items[2 * i].u;
        if (id >= np || seen[id]) throw gob::Error("gob index: path ids of a graph are not 0..n-1");
        seen[id] = 1;
        g.path_names[id] = paths->items[2 * i + 1].s;
    }
    if (lengths) {
        if (!lengths->packed) throw gob::Error("gob index: Lengths is not an integer map");
        for (size_t i = 0; i + 1 < lengths->nums.size(); i += 2) {
            if (lengths->nums[i] >= np) throw gob::Error("gob index: Lengths key without a path");
            g.path_len[lengths->nums[i]] = checked_u32(lengths->nums[i + 1], "path length");
        }
    }
    const Value *nodes = gv.field("SortedNodes");
    const size_t nn = nodes ? nodes->items.size() : 0;
    g.nodes.resize(nn);
    for (size_t i = 0; i < nn; i++) {
        const Value &nv = nodes->items[i];
        Node &n = g.nodes[i];
        n.seg_id = checked_u32(nv.uint_field("SegmentID"), "segment id");
        if (const Value *s = nv.field("Sequence")) n.seq = s->s;
        if (const Value *e = nv.field("OutEdges"))
            for (uint64_t x : e->nums) n.out.push_back(checked_u32(x, "out edge"));
        if (const Value *f = nv.field("KmerFreq")) n.kmer_freq = f->as_double();
        std::unordered_map<uint64_t, uint64_t> pos;
        if (const Value *pm = nv.field("Position")) {
            if (!pm->packed) throw gob::Error("gob index: Position is not an integer map");
            for (size_t j = 0; j + 1 < pm->nums.size(); j += 2) pos[pm->nums[j]] = pm->nums[j + 1];
        }
        if (const Value *pi = nv.field("PathIDs"))
            for (uint64_t x : pi->nums) {
                if (x >= np) throw gob::Error("gob index: node path id without a path");
                auto it = pos.find(x);
                if (it == pos.end()) throw gob::Error("gob index: node without a Position for one of its PathIDs");
                n.path_ids.push_back((uint32_t)x);
                n.pos.push_back(checked_u32(it->second, "node position"));
            }
    }
}

} // namespace

extern "C" {

// decodes every top-level value of a gob stream into JSON text (maps as [[k,v],...], absent fields omitted)
int groot_gob_to_json(const uint8_t *data, uint64_t n, char *out, uint64_t cap, uint64_t *needed)
{
    if (!data || !needed) return set_error(GROOT_E_INVALID, "null argument");
    try {
        gob::Decoder dec(data, (size_t)n);
        std::string text = "[";
        while (!dec.at_end()) {
            const Value v = dec.next();
            if (text.size() > 1) text += ',';
            gob::to_json(v, dec, text);
        }
        text += ']';
        *needed = text.size() + 1;
        if (out && cap >= text.size() + 1) memcpy(out, text.c_str(), text.size() + 1);
        return GROOT_OK;
    } catch (const std::exception &e) {
        return set_error(GROOT_E_FORMAT, "%s", e.what());
    }
}

// index.Load of the reference (cmd/align.go:93-107): <dir>/groot.gg + <dir>/groot.lshe -> flat index
int groot_index_load_gob(const char *gg_path, const char *lshe_path, groot_index **out)
{
    if (!gg_path || !lshe_path || !out) return set_error(GROOT_E_INVALID, "null argument");
    try {
        groot_index_params prm;
        groot_index_params_default(&prm);
        std::vector<Graph> graphs;
        {
            Blob gg;
            if (int rc = gg.load(gg_path)) return rc;
            gob::Decoder dec(gg.bytes.data(), gg.bytes.size());
            const Value info = dec.next();
            if (info.kind != gob::K_STRUCT) throw gob::Error("groot.gg does not hold a struct");
            prm.kmer_size = checked_u32(info.uint_field("KmerSize"), "KmerSize");
            prm.sketch_size = checked_u32(info.uint_field("SketchSize"), "SketchSize");
            prm.window_size = checked_u32(info.uint_field("WindowSize"), "WindowSize");
            prm.num_part = checked_u32(info.uint_field("NumPart"), "NumPart");
            prm.max_k = checked_u32(info.uint_field("MaxK"), "MaxK");
            prm.max_sketch_span = checked_u32(info.uint_field("MaxSketchSpan"), "MaxSketchSpan");
            const Value *store = info.field("Store");
            if (!store || store->count() == 0) throw gob::Error("groot.gg holds no graphs");
            const size_t ng = store->count();
            graphs.resize(ng);
            std::vector<uint8_t> seen(ng, 0);
            for (size_t i = 0; i < ng; i++) {
                const uint64_t key = store->items[2 * i].u;
                // graph ids are the MSA file iterator (src/pipeline/index.go:46-55): dense from 0
                if (key >= ng || seen[key]) throw gob::Error("groot.gg: graph ids are not 0..n-1");
                seen[key] = 1;
                graph_from_value(store->items[2 * i + 1], (uint32_t)key, graphs[key]);
            }
        }
        uint64_t num_window_kmers = 0;
        {
            Blob db;
            if (int rc = db.load(lshe_path)) return rc;
            std::vector<std::vector<LoadedWindow>> per_graph(graphs.size());
            gob::MapHook hook = [&](const std::string &field, Value &key, Value &val) -> bool {
                if (field != "WindowLookup") return false;
                if (key.kind != gob::K_STRING || val.kind != gob::K_STRUCT) throw gob::Error("groot.lshe: WindowLookup is not map[string]Key");
                LoadedWindow lw;
                if (!parse_lookup_index(key.s, lw.list_index)) throw gob::Error("groot.lshe: unexpected window key " + key.s);
                Window &w = lw.w;
                w.graph = checked_u32(val.uint_field("GraphID"), "GraphID");
                w.node_seg = checked_u32(val.uint_field("Node"), "Node");
                w.offset = checked_u32(val.uint_field("OffSet"), "OffSet");
                w.merge_span = checked_u32(val.uint_field("MergeSpan"), "MergeSpan");
                if (w.graph >= graphs.size()) throw gob::Error("groot.lshe: window of an unknown graph");
                if (val.uint_field("WindowSize", prm.window_size) != prm.window_size)
                    throw gob::Error("groot.lshe: window size differs from groot.gg");
                if (const Value *cn = val.field("ContainedNodes")) {
                    if (!cn->packed) throw gob::Error("groot.lshe: ContainedNodes is not a numeric map");
                    for (size_t j = 0; j + 1 < cn->nums.size(); j += 2) {
                        double d; memcpy(&d, &cn->nums[j + 1], 8);
                        if (!(d >= 0) || d > 4294967295.0 || d != (double)(uint32_t)d)
                            throw gob::Error("groot.lshe: ContainedNodes count is not a small integer");
                        w.contained.emplace_back(checked_u32(cn->nums[j], "contained node"), (uint32_t)d);
                    }
                    std::sort(w.contained.begin(), w.contained.end());
                }
                if (const Value *r = val.field("Ref"))
                    for (uint64_t x : r->nums) w.ref.push_back(checked_u32(x, "Ref"));
                if (const Value *s = val.field("Sketch")) w.sketch = s->nums;
                if (w.sketch.size() != prm.sketch_size) throw gob::Error("groot.lshe: sketch length differs from SketchSize");
                per_graph[w.graph].push_back(std::move(lw));
                return true;
            };
            gob::Decoder dec(db.bytes.data(), db.bytes.size());
            const Value ci = dec.next(&hook);
            if (ci.kind != gob::K_STRUCT) throw gob::Error("groot.lshe does not hold a struct");
            num_window_kmers = ci.uint_field("NumWindowKmers");
            if (ci.uint_field("SketchSize", prm.sketch_size) != prm.sketch_size || ci.uint_field("MaxK", prm.max_k) != prm.max_k ||
                ci.uint_field("NumPart", prm.num_part) != prm.num_part)
                throw gob::Error("groot.lshe: LSH parameters differ from groot.gg");
            size_t total = 0;
            for (size_t g = 0; g < graphs.size(); g++) {
                auto &v = per_graph[g];
                std::sort(v.begin(), v.end(), [](const LoadedWindow &a, const LoadedWindow &b) {
                    if (a.w.node_seg != b.w.node_seg) return a.w.node_seg < b.w.node_seg;
                    if (a.w.offset != b.w.offset) return a.w.offset < b.w.offset;
                    return a.list_index < b.list_index;
                });
                for (auto &lw : v) graphs[g].windows.push_back(std::move(lw.w));
                total += v.size();
            }
            if (!total) throw gob::Error("groot.lshe holds no windows");
        }
        if (int rc = check_index_params(&prm)) return rc;
        if (num_window_kmers != (uint64_t)prm.window_size - prm.kmer_size + 1)
            throw gob::Error("groot.lshe: NumWindowKmers is not WindowSize-KmerSize+1");
        if (int rc = flatten_graphs(graphs, prm, out)) return rc;
        if (int rc = groot_index_view_check(&(*out)->v)) {      // same pass as after groot_index_load
            groot_index_free(*out);
            *out = nullptr;
            return rc;
        }
        return GROOT_OK;
    } catch (const std::out_of_range &) {
        return set_error(GROOT_E_FORMAT, "gob index: a window or edge refers to a segment that is not in its graph");
    } catch (const std::exception &e) {
        return set_error(GROOT_E_FORMAT, "%s", e.what());
    }
}

// index.Dump + SaveDB of the reference (cmd/index.go:96-106,130-131; src/pipeline/runtime.go:64-72; src/lshe/lshe.go:71-92):
// <dir>/groot.gg and <dir>/groot.lshe with the fields `groot index` sets, so that the reference's own `groot align` /
// `groot haplotype` can load an index built here.  Maps are written in ascending key / window order (Go writes them in
// random order; a decoder does not care).
int groot_index_save_gob(const groot_index *idx, const char *dir, uint32_t max_sketch_span)
{
    using gob::GType;
    if (!idx || !dir) return set_error(GROOT_E_INVALID, "null argument");
    const groot_index_view &v = idx->v;
    auto basic = [](int id) { GType t; t.kind = GType::BASIC; t.basic = id; return t; };
    const GType BOOL = basic(1), INT = basic(2), UINT = basic(3), FLOAT = basic(4), BYTES = basic(5), STRING = basic(6);
    auto slice = [](const char *n, const GType *e) { GType t; t.kind = GType::SLICE; t.name = n; t.elem = e; return t; };
    auto map = [](const char *n, const GType *k, const GType *e) { GType t; t.kind = GType::MAP; t.name = n; t.key = k; t.elem = e; return t; };
    auto strct = [](const char *n, std::vector<std::pair<std::string, const GType *>> f) { GType t; t.kind = GType::STRUCT; t.name = n; t.fields = std::move(f); return t; };
    try {
        {   // ---- groot.gg: pipeline.Info ----
            const GType nodes_t = slice("Nodes", &UINT), u32s_t = slice("[]uint32", &UINT), pos_t = map("map[int]int", &INT, &INT);
            const GType node_t = strct("GrootGraphNode", {{"SegmentID", &UINT}, {"SegmentLength", &FLOAT}, {"Sequence", &BYTES}, {"OutEdges", &nodes_t},
                                                          {"PathIDs", &u32s_t}, {"Position", &pos_t}, {"KmerFreq", &FLOAT}, {"Marked", &BOOL}});
            const GType sorted_t = slice("[]*graph.GrootGraphNode", &node_t), paths_t = map("map[uint32][]uint8", &UINT, &BYTES),
                        lengths_t = map("map[uint32]int", &UINT, &INT), lookup_t = map("map[uint64]int", &UINT, &INT);
            const GType graph_t = strct("GrootGraph", {{"GrootVersion", &STRING}, {"GraphID", &UINT}, {"SortedNodes", &sorted_t}, {"Paths", &paths_t},
                                                       {"Lengths", &lengths_t}, {"NodeLookup", &lookup_t}, {"Masked", &BOOL}, {"KmerTotal", &UINT},
                                                       {"EMiterations", &INT}});
            const GType align_t = strct("AlignCmd", {{"Fasta", &BOOL}, {"BloomFilter", &BOOL}, {"MinKmerCoverage", &FLOAT}, {"BAMout", &STRING},
                                                     {"NoExactAlign", &BOOL}});
            const GType haplo_t = strct("HaploCmd", {{"Cutoff", &FLOAT}, {"MinIterations", &INT}, {"MaxIterations", &INT}, {"TotalKmers", &INT},
                                                     {"HaploDir", &STRING}});
            const GType store_t = map("Store", &UINT, &graph_t);
            const GType info_t = strct("Info", {{"Version", &STRING}, {"NumProc", &INT}, {"Profiling", &BOOL}, {"KmerSize", &INT}, {"SketchSize", &INT},
                                                {"WindowSize", &INT}, {"NumPart", &INT}, {"MaxK", &INT}, {"MaxSketchSpan", &INT},
                                                {"ContainmentThreshold", &FLOAT}, {"IndexDir", &STRING}, {"Store", &store_t}, {"Sketch", &align_t},
                                                {"Haplotype", &haplo_t}});
            std::string b;
            gob::StructWriter info(b);
            const char *ver = groot_host_version();
            info.str(0, ver, strlen(ver));
            info.i64(3, v.kmer_size); info.i64(4, v.sketch_size); info.i64(5, v.window_size); info.i64(6, v.num_part); info.i64(7, v.max_k);
            info.i64(8, max_sketch_span);
            info.str(10, dir, strlen(dir));
            info.field(11);
            gob::put_uint(b, v.n_graphs);
            for (uint32_t g = 0; g < v.n_graphs; g++) {
                gob::put_uint(b, g);
                gob::StructWriter gw(b);
                gw.u(1, g);
                const uint32_t n0 = v.graph_node_off[g], n1 = v.graph_node_off[g + 1], p0 = v.graph_path_off[g], p1 = v.graph_path_off[g + 1];
                if (n1 > n0) {
                    gw.field(2);
                    gob::put_uint(b, n1 - n0);
                    for (uint32_t n = n0; n < n1; n++) {
                        gob::StructWriter nw(b);
                        const uint32_t s0 = v.node_seq_off[n], s1 = v.node_seq_off[n + 1];
                        nw.u(0, v.node_seg_id[n]);
                        nw.f(1, (double)(s1 - s0));
                        nw.str(2, v.bases + s0, s1 - s0);
                        const uint32_t e0 = v.node_edge_off[n], e1 = v.node_edge_off[n + 1];
                        if (e1 > e0) { nw.field(3); gob::put_uint(b, e1 - e0); for (uint32_t e = e0; e < e1; e++) gob::put_uint(b, v.node_seg_id[v.edges[e]]); }
                        const uint32_t q0 = v.node_np_off[n], q1 = v.node_np_off[n + 1];
                        if (q1 > q0) { nw.field(4); gob::put_uint(b, q1 - q0); for (uint32_t q = q0; q < q1; q++) gob::put_uint(b, v.np_path[q]); }
                        nw.field(5);
                        gob::put_uint(b, q1 - q0);
                        for (uint32_t q = q0; q < q1; q++) { gob::put_int(b, v.np_path[q]); gob::put_int(b, v.np_pos[q]); }
                        nw.end();
                    }
                }
                gw.field(3);
                gob::put_uint(b, p1 - p0);
                for (uint32_t p = p0; p < p1; p++) {
                    gob::put_uint(b, p - p0);
                    gob::put_bytes(b, v.path_names + v.path_name_off[p], v.path_name_off[p + 1] - v.path_name_off[p]);
                }
                gw.field(4);
                gob::put_uint(b, p1 - p0);
                for (uint32_t p = p0; p < p1; p++) { gob::put_uint(b, p - p0); gob::put_int(b, v.path_len[p]); }
                gw.field(5);
                gob::put_uint(b, n1 - n0);
                for (uint32_t n = n0; n < n1; n++) { gob::put_uint(b, v.node_seg_id[n]); gob::put_int(b, n - n0); }
                gw.u(6, v.graph_masked[g] ? 1 : 0);
                gw.end();
            }
            info.field(12); b.push_back(0);      // Sketch AlignCmd{}: struct-typed fields are always sent
            info.field(13); b.push_back(0);      // Haplotype HaploCmd{}
            info.end();
            gob::Encoder enc;
            enc.message(&info_t, b);
            std::ofstream f(std::string(dir) + "/groot.gg", std::ios::binary);
            if (!f || !f.write(enc.out.data(), (std::streamsize)enc.out.size())) return set_error(GROOT_E_IO, "cannot write %s/groot.gg", dir);
        }
        {   // ---- groot.lshe: lshe.ContainmentIndex ----
            const GType cn_t = map("map[uint64]float64", &UINT, &FLOAT), ref_t = slice("[]uint32", &UINT), sk_t = slice("[]uint64", &UINT);
            const GType key_t = strct("Key", {{"GraphID", &UINT}, {"Node", &UINT}, {"OffSet", &UINT}, {"ContainedNodes", &cn_t}, {"Ref", &ref_t},
                                              {"RC", &BOOL}, {"Sketch", &sk_t}, {"Freq", &FLOAT}, {"MergeSpan", &UINT}, {"WindowSize", &UINT}});
            const GType look_t = map("map[string]lshe.Key", &STRING, &key_t);
            const GType ci_t = strct("ContainmentIndex", {{"NumPart", &INT}, {"MaxK", &INT}, {"NumWindowKmers", &INT}, {"SketchSize", &INT},
                                                          {"WindowLookup", &look_t}});
            std::string b;
            gob::StructWriter ci(b);
            ci.i64(0, v.num_part); ci.i64(1, v.max_k); ci.i64(2, v.num_window_kmers); ci.i64(3, v.sketch_size);
            ci.field(4);
            gob::put_uint(b, v.n_windows);
            uint32_t dup = 0;
            for (uint32_t w = 0; w < v.n_windows; w++) {
                const uint32_t g = v.win_graph[w], node = v.node_seg_id[v.win_node[w]], off = v.win_offset[w];
                // "g%dn%do%d-%d" (src/pipeline/index.go:199): windows sharing a start position are numbered in window order
                dup = (w && v.win_graph[w - 1] == g && v.win_node[w - 1] == v.win_node[w] && v.win_offset[w - 1] == off) ? dup + 1 : 0;
                char name[96];
                const int nl = snprintf(name, sizeof name, "g%un%uo%u-%u", g, node, off, dup);
                gob::put_bytes(b, name, (size_t)nl);
                gob::StructWriter kw(b);
                kw.u(0, g); kw.u(1, node); kw.u(2, off);
                kw.field(3);
                gob::put_uint(b, v.win_cn_off[w + 1] - v.win_cn_off[w]);
                for (uint32_t c = v.win_cn_off[w]; c < v.win_cn_off[w + 1]; c++) { gob::put_uint(b, v.node_seg_id[v.cn_node[c]]); gob::put_float(b, (double)v.cn_count[c]); }
                if (v.win_ref_off[w + 1] > v.win_ref_off[w]) {
                    kw.field(4);
                    gob::put_uint(b, v.win_ref_off[w + 1] - v.win_ref_off[w]);
                    for (uint32_t r = v.win_ref_off[w]; r < v.win_ref_off[w + 1]; r++) gob::put_uint(b, v.win_ref[r]);
                }
                kw.field(6);
                gob::put_uint(b, v.sketch_size);
                for (uint32_t i = 0; i < v.sketch_size; i++) gob::put_uint(b, v.win_sketch[(size_t)w * v.sketch_size + i]);
                kw.u(8, v.win_merge_span[w]);
                kw.u(9, v.window_size);
                kw.end();
            }
            ci.end();
            gob::Encoder enc;
            enc.message(&ci_t, b);
            std::ofstream f(std::string(dir) + "/groot.lshe", std::ios::binary);
            if (!f || !f.write(enc.out.data(), (std::streamsize)enc.out.size())) return set_error(GROOT_E_IO, "cannot write %s/groot.lshe", dir);
        }
        return GROOT_OK;
    } catch (const std::exception &e) {
        return set_error(GROOT_E_FORMAT, "%s", e.what());
    }
}

} // extern "C"
