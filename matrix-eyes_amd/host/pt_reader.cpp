#include "pt_reader.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <map>
#include <memory>

namespace matrix_eyes {

namespace {

uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

struct Member {
    const uint8_t* data;
    uint64_t size;
};

// ZIP central directory -> name -> stored bytes.  torch.save never compresses.  Every offset comes from the
// file: each read is checked against the mapped size first, in subtraction form so that a huge offset cannot
// wrap the comparison.
std::map<std::string, Member> zip_members(const uint8_t* f, size_t n, const std::string& path) {
    auto fits = [n](uint64_t off, uint64_t len) { return len <= n && off <= n - len; };
    auto need = [&](uint64_t off, uint64_t len, const char* what) {
        if (!fits(off, len)) throw CheckpointError(path + ": " + what);
    };
    if (n < 22) throw CheckpointError(path + ": not a zip archive");
    size_t eocd = n - 22;
    const size_t stop = n > 22 + 65535 ? n - 22 - 65535 : 0;
    while (le32(f + eocd) != 0x06054b50) {
        if (eocd == stop) throw CheckpointError(path + ": no zip end-of-central-directory record (legacy torch.save format?)");
        --eocd;
    }
    uint64_t count = le16(f + eocd + 10), cd_off = le32(f + eocd + 16);
    if (eocd >= 20 && le32(f + eocd - 20) == 0x07064b50) {  // zip64 locator
        const uint64_t e64 = le64(f + eocd - 20 + 8);
        need(e64, 56, "bad zip64 record");
        if (le32(f + e64) != 0x06064b50) throw CheckpointError(path + ": bad zip64 record");
        count = le64(f + e64 + 32), cd_off = le64(f + e64 + 48);
    }
    if (count > n / 46) throw CheckpointError(path + ": bad zip central directory");
    std::map<std::string, Member> out;
    uint64_t p = cd_off;
    for (uint64_t i = 0; i < count; ++i) {
        need(p, 46, "bad zip central directory");
        if (le32(f + p) != 0x02014b50) throw CheckpointError(path + ": bad zip central directory");
        const uint16_t method = le16(f + p + 10), nlen = le16(f + p + 28), xlen = le16(f + p + 30), clen = le16(f + p + 32);
        uint64_t csize = le32(f + p + 20), usize = le32(f + p + 24), lho = le32(f + p + 42);
        need(p + 46, (uint64_t)nlen + xlen + clen, "truncated zip central directory");
        const std::string name((const char*)f + p + 46, nlen);
        const uint8_t* x = f + p + 46 + nlen;  // zip64 extra field
        for (uint32_t o = 0; o + 4 <= xlen;) {
            const uint16_t id = le16(x + o), sz = le16(x + o + 2);
            if (o + 4 + (uint32_t)sz > xlen) throw CheckpointError(path + ": bad zip extra field");
            if (id == 1) {
                uint32_t q = o + 4;
                const uint32_t end = o + 4 + sz;
                auto take64 = [&](uint64_t& dst) {
                    if (q + 8 > end) throw CheckpointError(path + ": bad zip64 extra field");
                    dst = le64(x + q), q += 8;
                };
                if (usize == 0xffffffffu) take64(usize);
                if (csize == 0xffffffffu) take64(csize);
                if (lho == 0xffffffffu) take64(lho);
            }
            o += 4 + sz;
        }
        if (method != 0) throw CheckpointError(path + ": member " + name + " is compressed");
        need(lho, 30, "bad zip local header");
        if (le32(f + lho) != 0x04034b50) throw CheckpointError(path + ": bad zip local header");
        const uint64_t start = lho + 30 + le16(f + lho + 26) + le16(f + lho + 28);
        if (!fits(start, usize)) throw CheckpointError(path + ": member " + name + " runs past the end of the file");
        out[name] = Member{f + start, usize};
        p += 46 + (uint64_t)nlen + xlen + clen;
    }
    return out;
}

// ---- pickle ----------------------------------------------------------------------------------------
struct Value;
typedef std::shared_ptr<Value> Ref;
struct Value {
    enum Kind { NONE, INT, BOOL, STR, TUPLE, LIST, DICT, GLOBAL, STORAGE, TENSOR, OPAQUE, MARK } kind = NONE;
    int64_t i = 0;
    std::string s, s2;                            // STR; GLOBAL module / name; STORAGE key / dtype
    std::vector<Ref> items;                       // TUPLE, LIST
    std::vector<std::pair<Ref, Ref>> dict;        // DICT (insertion order)
    Ref storage;                                  // TENSOR
    int64_t offset = 0;
    std::vector<int64_t> dims, strides;
};
Ref make(Value::Kind k) {
    Ref r = std::make_shared<Value>();
    r->kind = k;
    return r;
}

const char* storage_dtype(const std::string& cls) {
    if (cls == "HalfStorage") return "f16";
    if (cls == "BFloat16Storage") return "bf16";
    if (cls == "FloatStorage") return "f32";
    if (cls == "DoubleStorage") return "f64";
    if (cls == "LongStorage") return "i64";
    if (cls == "IntStorage") return "i32";
    if (cls == "ByteStorage") return "u8";
    if (cls == "BoolStorage") return "bool";
    return nullptr;
}
size_t dtype_size(const std::string& d) {
    if (d == "f16" || d == "bf16") return 2;
    if (d == "f32" || d == "i32") return 4;
    if (d == "f64" || d == "i64") return 8;
    return 1;
}

class Unpickler {
public:
    Unpickler(const uint8_t* p, size_t n, const std::string& path) : p_(p), n_(n), path_(path) {}

    Ref run() {
        for (;;) {
            const uint8_t op = u8();
            switch (op) {
                case 0x80: u8(); break;                                  // PROTO
                case 0x95: take(8); break;                               // FRAME
                case '}': push(make(Value::DICT)); break;
                case ']': push(make(Value::LIST)); break;
                case ')': push(make(Value::TUPLE)); break;
                case '(': push(make(Value::MARK)); break;
                case 'N': push(make(Value::NONE)); break;
                case 0x88: case 0x89: { Ref v = make(Value::BOOL); v->i = op == 0x88; push(v); break; }
                case 'J': { Ref v = make(Value::INT); v->i = (int32_t)le32(take(4)); push(v); break; }
                case 'K': { Ref v = make(Value::INT); v->i = u8(); push(v); break; }
                case 'M': { Ref v = make(Value::INT); v->i = le16(take(2)); push(v); break; }
                case 0x8a: {                                             // LONG1
                    const uint8_t len = u8();
                    const uint8_t* b = take(len);
                    int64_t x = 0;
                    for (int k = 0; k < len && k < 8; ++k) x |= (int64_t)b[k] << (8 * k);
                    if (len && len < 8 && (b[len - 1] & 0x80)) x |= (int64_t)(~0ull << (8 * len));
                    Ref v = make(Value::INT); v->i = x; push(v); break;
                }
                case 'G': take(8); push(make(Value::OPAQUE)); break;     // BINFLOAT
                case 'X': { const uint32_t len = le32(take(4)); push(str(len)); break; }
                case 0x8c: { const uint8_t len = u8(); push(str(len)); break; }
                case 'T': { const uint32_t len = le32(take(4)); push(str(len)); break; }
                case 'U': { const uint8_t len = u8(); push(str(len)); break; }
                case 'c': {                                              // GLOBAL
                    Ref g = make(Value::GLOBAL);
                    g->s = line(), g->s2 = line();
                    push(g); break;
                }
                case 0x93: {                                             // STACK_GLOBAL
                    Ref name = pop(), mod = pop();
                    if (name->kind != Value::STR || mod->kind != Value::STR) bad("STACK_GLOBAL without two strings");
                    Ref g = make(Value::GLOBAL);
                    g->s = mod->s, g->s2 = name->s;
                    push(g); break;
                }
                case 'q': memo_[u8()] = top(); break;
                case 'r': memo_[le32(take(4))] = top(); break;
                case 0x94: memo_[(uint32_t)memo_.size()] = top(); break;  // MEMOIZE
                case 'h': push(get(u8())); break;
                case 'j': push(get(le32(take(4)))); break;
                case 't': { Ref t = make(Value::TUPLE); t->items = pop_to_mark(); push(t); break; }
                case 0x85: case 0x86: case 0x87: {
                    Ref t = make(Value::TUPLE);
                    const int k = op - 0x84;
                    t->items.resize(k);
                    for (int j = k - 1; j >= 0; --j) t->items[j] = pop();
                    push(t); break;
                }
                // a container opcode applied to anything else (a damaged file) must not write into it
                case 'a': { Ref v = pop(); list_top()->items.push_back(v); break; }
                case 'e': { std::vector<Ref> v = pop_to_mark(); for (Ref& x : v) list_top()->items.push_back(x); break; }
                case 's': { Ref v = pop(), k = pop(); dict_top()->dict.emplace_back(k, v); break; }
                case 'u': {
                    std::vector<Ref> kv = pop_to_mark();
                    Ref& d = dict_top();
                    for (size_t j = 0; j + 1 < kv.size(); j += 2) d->dict.emplace_back(kv[j], kv[j + 1]);
                    break;
                }
                case 'Q': push(persistent(pop())); break;                // BINPERSID
                case 'R': { Ref args = pop(), fn = pop(); push(reduce(fn, args)); break; }
                case 0x81: { Ref args = pop(), cls = pop(); push(reduce(cls, args)); break; }  // NEWOBJ
                case 'b': {                                              // BUILD: state onto an object
                    Ref state = pop();
                    if (top()->kind == Value::DICT && state->kind == Value::DICT)
                        for (auto& kv : state->dict) top()->dict.push_back(kv);
                    break;
                }
                case '.': return pop();
                default:
                    throw CheckpointError(path_ + ": pickle opcode 0x" + hex(op) + " is not one a state_dict uses");
            }
        }
    }

private:
    static std::string hex(uint8_t v) {
        const char* d = "0123456789abcdef";
        return std::string() + d[v >> 4] + d[v & 15];
    }
    uint8_t u8() { return *take(1); }
    const uint8_t* take(size_t k) {
        if (pos_ + k > n_) throw CheckpointError(path_ + ": truncated pickle");
        const uint8_t* r = p_ + pos_;
        pos_ += k;
        return r;
    }
    std::string line() {
        std::string s;
        for (uint8_t c; (c = u8()) != '\n';) s.push_back((char)c);
        return s;
    }
    Ref str(size_t len) {
        Ref v = make(Value::STR);
        v->s.assign((const char*)take(len), len);
        return v;
    }
    void push(Ref v) { stack_.push_back(std::move(v)); }
    Ref pop() {
        if (stack_.empty()) throw CheckpointError(path_ + ": pickle stack underflow");
        Ref v = stack_.back();
        stack_.pop_back();
        return v;
    }
    Ref& top() {
        if (stack_.empty()) throw CheckpointError(path_ + ": pickle stack underflow");
        return stack_.back();
    }
    [[noreturn]] void bad(const char* what) { throw CheckpointError(path_ + ": malformed pickle: " + what); }
    // objects of classes this reader does not model are OPAQUE: items stored into them are dropped (the
    // Value keeps them in vectors nothing reads); anything else is a damaged stream
    Ref& list_top() {
        if (top()->kind != Value::LIST && top()->kind != Value::OPAQUE) bad("APPEND(S) onto something that is not a list");
        return top();
    }
    Ref& dict_top() {
        if (top()->kind != Value::DICT && top()->kind != Value::OPAQUE) bad("SETITEM(S) onto something that is not a dict");
        return top();
    }
    Ref get(uint32_t k) {
        auto it = memo_.find(k);
        if (it == memo_.end()) throw CheckpointError(path_ + ": pickle memo miss");
        return it->second;
    }
    std::vector<Ref> pop_to_mark() {
        std::vector<Ref> v;
        for (;;) {
            Ref x = pop();
            if (x->kind == Value::MARK) break;
            v.push_back(x);
        }
        return std::vector<Ref>(v.rbegin(), v.rend());
    }
    // ('storage', <class torch.XStorage>, key, location, numel)
    Ref persistent(const Ref& pid) {
        if (pid->kind != Value::TUPLE || pid->items.size() < 3 || pid->items[0]->kind != Value::STR ||
            pid->items[0]->s != "storage" || pid->items[1]->kind != Value::GLOBAL || pid->items[2]->kind != Value::STR)
            throw CheckpointError(path_ + ": unknown persistent id");
        const char* dt = storage_dtype(pid->items[1]->s2);
        if (!dt) throw CheckpointError(path_ + ": storage class " + pid->items[1]->s2 + " is not supported");
        Ref st = make(Value::STORAGE);
        st->s = pid->items[2]->s, st->s2 = dt;
        return st;
    }
    std::vector<int64_t> ints(const Ref& t) {
        if (t->kind != Value::TUPLE) bad("tensor size / stride is not a tuple");
        std::vector<int64_t> v;
        for (const Ref& x : t->items) {
            if (x->kind != Value::INT) bad("tensor size / stride entry is not an integer");
            v.push_back(x->i);
        }
        return v;
    }
    Ref reduce(const Ref& fn, const Ref& args) {
        if (fn->kind == Value::GLOBAL && args->kind == Value::TUPLE) {
            if (fn->s == "collections" && fn->s2 == "OrderedDict") return make(Value::DICT);
            if (fn->s == "torch._utils" && fn->s2 == "_rebuild_tensor_v2" && args->items.size() >= 4) {
                Ref t = make(Value::TENSOR);
                t->storage = args->items[0];
                if (args->items[1]->kind != Value::INT) bad("tensor storage offset is not an integer");
                t->offset = args->items[1]->i;
                t->dims = ints(args->items[2]), t->strides = ints(args->items[3]);
                return t;
            }
            if (fn->s == "torch._utils" && fn->s2 == "_rebuild_parameter" && !args->items.empty()) return args->items[0];
        }
        return make(Value::OPAQUE);
    }

    const uint8_t* p_;
    size_t n_, pos_ = 0;
    std::string path_;
    std::vector<Ref> stack_;
    std::map<uint32_t, Ref> memo_;
};

void collect(const Ref& v, const std::string& prefix, const std::map<std::string, Member>& members,
             const std::string& root, const std::string& path, std::vector<PtTensor>& out) {
    if (v->kind != Value::DICT) return;
    for (const auto& kv : v->dict) {
        if (kv.first->kind != Value::STR) continue;
        const std::string name = prefix + kv.first->s;
        const Ref& t = kv.second;
        if (t->kind == Value::DICT) {  // {"state_dict": {...}} style wrappers
            collect(t, "", members, root, path, out);
            continue;
        }
        if (t->kind != Value::TENSOR || !t->storage || t->storage->kind != Value::STORAGE) continue;
        if (t->strides.size() != t->dims.size() || t->offset < 0)
            throw CheckpointError(path + ": tensor " + name + " has a malformed size / stride / offset");
        auto it = members.find(root + "data/" + t->storage->s);
        if (it == members.end()) throw CheckpointError(path + ": storage " + t->storage->s + " of " + name + " is missing");
        const uint64_t esz = dtype_size(t->storage->s2), cap = it->second.size / esz;  // elements the storage holds
        uint64_t expect = 1;
        for (size_t d = t->dims.size(); d-- > 0;) {
            if (t->dims[d] < 0) throw CheckpointError(path + ": tensor " + name + " has a negative dimension");
            if (t->dims[d] != 1 && t->strides[d] != (int64_t)expect)
                throw CheckpointError(path + ": tensor " + name + " is not contiguous");
            // the product stays <= cap (no overflow): a larger tensor cannot lie inside its storage anyway
            if (t->dims[d] != 0 && expect > cap / (uint64_t)t->dims[d])
                throw CheckpointError(path + ": tensor " + name + " runs past its storage");
            expect *= (uint64_t)t->dims[d];
        }
        if ((uint64_t)t->offset > cap || expect > cap - (uint64_t)t->offset)
            throw CheckpointError(path + ": tensor " + name + " runs past its storage");
        const size_t nbytes = (size_t)(expect * esz), off = (size_t)((uint64_t)t->offset * esz);
        PtTensor e;
        e.name = name, e.dtype = t->storage->s2, e.dims = t->dims;
        e.data = it->second.data + off, e.nbytes = nbytes;
        out.push_back(std::move(e));
    }
}

}  // namespace

PtFile::PtFile(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw CheckpointError("cannot open checkpoint " + path + ": " + std::strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) {
        ::close(fd);
        throw CheckpointError("cannot stat checkpoint " + path);
    }
    size_ = (size_t)st.st_size;
    map_ = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED) {
        map_ = nullptr;
        throw CheckpointError("cannot map checkpoint " + path);
    }
    try {
        const uint8_t* f = (const uint8_t*)map_;
        const std::map<std::string, Member> members = zip_members(f, size_, path);
        std::string root;
        const Member* pkl = nullptr;
        for (const auto& m : members) {
            const std::string& n = m.first;
            if (n.size() >= 8 && n.compare(n.size() - 8, 8, "data.pkl") == 0) {
                root = n.substr(0, n.size() - 8);
                pkl = &m.second;
                break;
            }
        }
        if (!pkl) throw CheckpointError(path + ": no data.pkl in the archive");
        Unpickler up(pkl->data, pkl->size, path);
        collect(up.run(), "", members, root, path, tensors_);
        if (tensors_.empty()) throw CheckpointError(path + ": no tensors found in the checkpoint");
    } catch (...) {
        munmap(map_, size_);
        map_ = nullptr;
        throw;
    }
}

PtFile::~PtFile() {
    if (map_) munmap(map_, size_);
}

}  // namespace matrix_eyes
