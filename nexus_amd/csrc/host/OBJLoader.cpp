// OBJLoader.cpp — binary glTF 2.0 and Wavefront .obj ingestion without Assimp (see nexus/OBJLoader.h).
// What the reference obtains through Assimp (/root/reference/Nexus/src/Assets/OBJLoader.cpp:8-239) is reproduced for the
// subset of the formats its demo assets use; the numbers (TRS decomposition, roughness mapping) are computed in double
// and stored as float, like the Python reader in nexus_amd/loaders.py that the tests compare against.
#include "nexus/OBJLoader.h"

#include "nexus/IMGLoader.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "nexus/Scene.h"

namespace nexus {

namespace {

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("OBJLoader: " + what); }

// ---- a small JSON reader (objects, arrays, strings, numbers, true / false / null) ----------------------------------
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const std::string& key) const
    {
        if (kind != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& at(const std::string& key) const
    {
        const Json* j = find(key);
        if (!j) fail("glTF JSON: missing key \"" + key + "\"");
        return *j;
    }
    const Json& at(size_t i) const
    {
        if (kind != Array || i >= arr.size()) fail("glTF JSON: array index out of range");
        return arr[i];
    }
    double number(const std::string& key, double dflt) const
    {
        const Json* j = find(key);
        return (j && j->kind == Number) ? j->num : dflt;
    }
    size_t size() const { return kind == Array ? arr.size() : 0; }
};

class JsonParser {
public:
    JsonParser(const char* p, size_t n) : m_P(p), m_End(p + n) {}
    Json parse()
    {
        Json j = value();
        skip();
        return j;
    }

private:
    const char* m_P;
    const char* m_End;
    void skip()
    {
        while (m_P < m_End && (*m_P == ' ' || *m_P == '\n' || *m_P == '\r' || *m_P == '\t')) m_P++;
    }
    char peek()
    {
        skip();
        if (m_P >= m_End) fail("glTF JSON: unexpected end");
        return *m_P;
    }
    void expect(char c)
    {
        if (peek() != c) fail(std::string("glTF JSON: expected '") + c + "'");
        m_P++;
    }
    Json value()
    {
        const char c = peek();
        Json j;
        if (c == '{') {
            j.kind = Json::Object;
            m_P++;
            if (peek() == '}') { m_P++; return j; }
            for (;;) {
                Json k = string();
                expect(':');
                j.obj.emplace_back(std::move(k.str), value());
                if (peek() == ',') { m_P++; continue; }
                expect('}');
                return j;
            }
        }
        if (c == '[') {
            j.kind = Json::Array;
            m_P++;
            if (peek() == ']') { m_P++; return j; }
            for (;;) {
                j.arr.push_back(value());
                if (peek() == ',') { m_P++; continue; }
                expect(']');
                return j;
            }
        }
        if (c == '"') return string();
        if (c == 't' || c == 'f' || c == 'n') {
            const char* words[3] = {"true", "false", "null"};
            for (int w = 0; w < 3; w++) {
                const size_t len = std::strlen(words[w]);
                if (static_cast<size_t>(m_End - m_P) >= len && std::strncmp(m_P, words[w], len) == 0) {
                    m_P += len;
                    if (w < 2) { j.kind = Json::Bool; j.b = (w == 0); }
                    return j;
                }
            }
            fail("glTF JSON: bad literal");
        }
        // number
        char* end = nullptr;
        const std::string tmp(m_P, std::min<size_t>(static_cast<size_t>(m_End - m_P), 64));
        const double v = std::strtod(tmp.c_str(), &end);
        if (end == tmp.c_str()) fail("glTF JSON: bad number");
        m_P += end - tmp.c_str();
        j.kind = Json::Number;
        j.num = v;
        return j;
    }
    Json string()
    {
        expect('"');
        Json j;
        j.kind = Json::String;
        while (m_P < m_End && *m_P != '"') {
            if (*m_P == '\\') {
                m_P++;
                if (m_P >= m_End) break;
                switch (*m_P) {
                case 'n': j.str += '\n'; break;
                case 't': j.str += '\t'; break;
                case 'r': j.str += '\r'; break;
                case 'b': j.str += '\b'; break;
                case 'f': j.str += '\f'; break;
                case 'u':  // names only: keep ASCII, replace the rest
                    if (m_End - m_P >= 5) {
                        const unsigned long cp = std::strtoul(std::string(m_P + 1, 4).c_str(), nullptr, 16);
                        j.str += cp < 128 ? static_cast<char>(cp) : '?';
                        m_P += 4;
                    }
                    break;
                default: j.str += *m_P; break;
                }
                m_P++;
            } else {
                j.str += *m_P++;
            }
        }
        if (m_P >= m_End) fail("glTF JSON: unterminated string");
        m_P++;
        return j;
    }
};

// ---- 4x4 double matrices, row major ----------------------------------------------------------------------------------
struct M4 {
    double m[4][4];
    static M4 identity()
    {
        M4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0;
        return r;
    }
    M4 operator*(const M4& o) const
    {
        M4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double s = 0.0;
                for (int k = 0; k < 4; k++) s += m[i][k] * o.m[k][j];
                r.m[i][j] = s;
            }
        return r;
    }
};

M4 node_local_matrix(const Json& node)
{
    M4 r = M4::identity();
    if (const Json* mat = node.find("matrix")) {  // glTF stores column major
        if (mat->size() != 16) fail("glTF node matrix must have 16 entries");
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r.m[i][j] = mat->at(static_cast<size_t>(j * 4 + i)).num;
        return r;
    }
    double q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1}, t[3] = {0, 0, 0};
    if (const Json* j = node.find("rotation"))
        for (size_t k = 0; k < 4 && k < j->size(); k++) q[k] = j->at(k).num;
    if (const Json* j = node.find("scale"))
        for (size_t k = 0; k < 3 && k < j->size(); k++) s[k] = j->at(k).num;
    if (const Json* j = node.find("translation"))
        for (size_t k = 0; k < 3 && k < j->size(); k++) t[k] = j->at(k).num;
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                            {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                            {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) r.m[i][j] = R[i][j] * s[j];
        r.m[i][3] = t[i];
    }
    return r;
}

// aiMatrix4x4::Decompose as the reference consumes it: position, Euler XYZ in degrees, scale
void decompose_trs(const M4& m, float3& pos, float3& rotDeg, float3& scale)
{
    double c[3][3];  // columns
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++) c[k][i] = m.m[i][k];
    double sc[3];
    for (int k = 0; k < 3; k++) sc[k] = std::sqrt(c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2]);
    const double det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) - m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                       m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
    if (det < 0)
        for (int k = 0; k < 3; k++) sc[k] = -sc[k];
    for (int k = 0; k < 3; k++)
        if (sc[k] != 0)
            for (int i = 0; i < 3; i++) c[k][i] /= sc[k];
    const double eps = 1e-10;
    const double ry = std::asin(-c[0][2]);
    const double cy = std::cos(ry);
    double rx, rz;
    if (std::fabs(cy) > eps) {
        rx = std::atan2(c[1][2], c[2][2]);
        rz = std::atan2(c[0][1], c[0][0]);
    } else {
        rx = 0.0;
        rz = std::atan2(-c[1][0], c[1][1]);
    }
    const double deg = 180.0 / 3.141592653589793238462643383279502884;
    pos = make_float3(static_cast<float>(m.m[0][3]), static_cast<float>(m.m[1][3]), static_cast<float>(m.m[2][3]));
    rotDeg = make_float3(static_cast<float>(rx * deg), static_cast<float>(ry * deg), static_cast<float>(rz * deg));
    scale = make_float3(static_cast<float>(sc[0]), static_cast<float>(sc[1]), static_cast<float>(sc[2]));
}

std::vector<unsigned char> read_file(const std::string& file)
{
    std::ifstream f(file, std::ios::binary);
    if (!f) fail("cannot open " + file);
    std::vector<unsigned char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return data;
}

uint32_t rd32(const std::vector<unsigned char>& d, size_t off)
{
    if (off + 4 > d.size()) fail("glb: truncated file");
    uint32_t v;
    std::memcpy(&v, d.data() + off, 4);
    return v;
}

void set_albedo_like(Material& m, const double rgb[3], double roughness, double ior)
{
    m.plastic.albedo[0] = static_cast<float>(rgb[0]);
    m.plastic.albedo[1] = static_cast<float>(rgb[1]);
    m.plastic.albedo[2] = static_cast<float>(rgb[2]);
    m.plastic.roughness = static_cast<float>(roughness);
    m.plastic.ior = static_cast<float>(ior);
}

// OBJLoader.cpp:71-163 through Assimp's glTF importer
Material gltf_material(const Json& m)
{
    static const Json empty;
    const Json* pbr = m.find("pbrMetallicRoughness");
    const Json* ext = m.find("extensions");
    double base[4] = {1, 1, 1, 1};
    if (pbr)
        if (const Json* b = pbr->find("baseColorFactor"))
            for (size_t k = 0; k < 4 && k < b->size(); k++) base[k] = b->at(k).num;
    const double rough = pbr ? pbr->number("roughnessFactor", 1.0) : 1.0;
    const double shininess = (1.0 - rough) * (1.0 - rough) * 1000.0;  // Assimp: roughness -> shininess
    double roughness = 1.0 - std::sqrt(shininess) / 31.62278;         // the reference maps it back
    roughness = std::min(1.0, std::max(0.0, roughness));
    auto ext_number = [&](const char* e, const char* key, double dflt) {
        if (!ext) return dflt;
        const Json* j = ext->find(e);
        return j ? j->number(key, dflt) : dflt;
    };
    const double ior = ext_number("KHR_materials_ior", "ior", 1.45);
    const double transmission = ext_number("KHR_materials_transmission", "transmissionFactor", 0.0);
    Material out;
    out.type = transmission > 0.0 ? NX_MAT_DIELECTRIC : NX_MAT_PLASTIC;
    set_albedo_like(out, base, roughness, ior);
    if (const Json* e = m.find("emissiveFactor"))
        for (size_t k = 0; k < 3 && k < e->size(); k++) out.emissive[k] = static_cast<float>(e->at(k).num);
    out.intensity = static_cast<float>(ext_number("KHR_materials_emissive_strength", "emissiveStrength", 1.0));
    out.opacity = static_cast<float>(base[3]);
    return out;
}

float3 face_normal(float3 p0, float3 p1, float3 p2)
{
    const float3 fn = cross(p1 - p0, p2 - p0);
    const float ln = std::sqrt(fn.x * fn.x + fn.y * fn.y + fn.z * fn.z);
    if (!(ln > 0.0f)) return make_float3(0.0f, 0.0f, 1.0f);
    const float d = std::max(ln, 1e-30f);
    return make_float3(fn.x / d, fn.y / d, fn.z / d);
}

struct Accessor {
    const unsigned char* base = nullptr;
    size_t stride = 0, count = 0;
    int componentType = 0, ncomp = 0;
    double get(size_t i, int c) const
    {
        const unsigned char* p = base + i * stride;
        switch (componentType) {
        case 5120: { int8_t v; std::memcpy(&v, p + c, 1); return v; }
        case 5121: { uint8_t v; std::memcpy(&v, p + c, 1); return v; }
        case 5122: { int16_t v; std::memcpy(&v, p + 2 * c, 2); return v; }
        case 5123: { uint16_t v; std::memcpy(&v, p + 2 * c, 2); return v; }
        case 5125: { uint32_t v; std::memcpy(&v, p + 4 * c, 4); return v; }
        case 5126: { float v; std::memcpy(&v, p + 4 * c, 4); return v; }
        default: fail("glb: unsupported accessor component type");
        }
    }
};

LoadedScene parse_glb(const std::string& file)
{
    const std::vector<unsigned char> data = read_file(file);
    if (data.size() < 20 || rd32(data, 0) != 0x46546C67u || rd32(data, 4) != 2u) fail(file + " is not a glTF 2.0 binary file");
    size_t off = 12;
    const unsigned char* jsonChunk = nullptr;
    size_t jsonLen = 0;
    const unsigned char* blob = nullptr;
    size_t blobLen = 0;
    while (off + 8 <= data.size()) {
        const uint32_t clen = rd32(data, off), ctype = rd32(data, off + 4);
        if (off + 8 + clen > data.size()) fail("glb: chunk runs past the end of the file");
        if (ctype == 0x4E4F534Au) { jsonChunk = data.data() + off + 8; jsonLen = clen; }
        else if (ctype == 0x004E4942u) { blob = data.data() + off + 8; blobLen = clen; }
        off += 8 + static_cast<size_t>(clen);
    }
    if (!jsonChunk) fail("glb without a JSON chunk");
    const Json doc = JsonParser(reinterpret_cast<const char*>(jsonChunk), jsonLen).parse();

    auto accessor = [&](size_t index) {
        const Json& a = doc.at("accessors").at(index);
        const Json& bv = doc.at("bufferViews").at(static_cast<size_t>(a.at("bufferView").num));
        Accessor acc;
        acc.componentType = static_cast<int>(a.at("componentType").num);
        const std::string& type = a.at("type").str;
        acc.ncomp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
        if (!acc.ncomp) fail("glb: unsupported accessor type " + type);
        const size_t csize = (acc.componentType == 5120 || acc.componentType == 5121) ? 1 : (acc.componentType == 5122 || acc.componentType == 5123) ? 2 : 4;
        const size_t start = static_cast<size_t>(bv.number("byteOffset", 0)) + static_cast<size_t>(a.number("byteOffset", 0));
        const size_t stride = static_cast<size_t>(bv.number("byteStride", 0));
        acc.stride = stride ? stride : csize * static_cast<size_t>(acc.ncomp);
        acc.count = static_cast<size_t>(a.at("count").num);
        if (!blob || (acc.count && start + (acc.count - 1) * acc.stride + csize * static_cast<size_t>(acc.ncomp) > blobLen)) fail("glb: accessor runs past the binary chunk");
        acc.base = blob + start;
        return acc;
    };

    LoadedScene out;
    // texture index -> decoded image, decoded once however many materials share it
    std::map<std::pair<size_t, int>, int> decoded;  // (glTF texture, kind) -> index into out.textures
    auto load_texture = [&](const Json& texRef, Texture::Type kind) -> int {
        const size_t ti = static_cast<size_t>(texRef.at("index").num);
        const auto key = std::make_pair(ti, kind == Texture::Type::DIFFUSE ? 0 : 1);
        const auto hit = decoded.find(key);
        if (hit != decoded.end()) return hit->second;
        int result = -1;
        try {
            const Json& tex = doc.at("textures").at(ti);
            const Json& img = doc.at("images").at(static_cast<size_t>(tex.at("source").num));
            Texture t;
            if (const Json* bvRef = img.find("bufferView")) {
                const Json& bv = doc.at("bufferViews").at(static_cast<size_t>(bvRef->num));
                const size_t start = static_cast<size_t>(bv.number("byteOffset", 0)), len = static_cast<size_t>(bv.at("byteLength").num);
                if (!blob || start + len > blobLen) fail("glb: image runs past the binary chunk");
                t = IMGLoader::LoadIMG(blob + start, len);
            } else if (const Json* uri = img.find("uri")) {
                if (uri->str.compare(0, 5, "data:") == 0) fail("glb: data: URIs are not supported for images");
                const size_t slash = file.find_last_of("/\\");
                t = IMGLoader::LoadIMG((slash == std::string::npos ? std::string() : file.substr(0, slash + 1)) + uri->str);
            } else {
                fail("glb: image without bufferView or uri");
            }
            t.type = kind;
            result = static_cast<int>(out.textures.size());
            out.textures.push_back(std::move(t));
        } catch (const std::exception& e) {
            out.warnings.push_back(std::string("texture ") + std::to_string(ti) + ": " + e.what());
        }
        decoded[key] = result;
        return result;
    };
    if (const Json* mats = doc.find("materials"))
        for (size_t i = 0; i < mats->size(); i++) {
            const Json& m = mats->at(i);
            out.materials.push_back(gltf_material(m));
            int dt = -1, et = -1;
            if (const Json* pbr = m.find("pbrMetallicRoughness"))
                if (const Json* t = pbr->find("baseColorTexture")) dt = load_texture(*t, Texture::Type::DIFFUSE);
            if (const Json* t = m.find("emissiveTexture")) et = load_texture(*t, Texture::Type::EMISSIVE);
            out.materialDiffuseTexture.push_back(dt);
            out.materialEmissiveTexture.push_back(et);
        }
    if (out.materials.empty()) {
        Material m;  // pod.make_material() defaults
        m.diffuse.albedo[0] = m.diffuse.albedo[1] = m.diffuse.albedo[2] = 0.8f;
        m.plastic.ior = 1.45f;
        out.materials.push_back(m);
        out.materialDiffuseTexture.push_back(-1);
        out.materialEmissiveTexture.push_back(-1);
    }

    // one mesh per primitive (what Assimp hands the reference, OBJLoader.cpp:165-181)
    std::map<std::pair<size_t, size_t>, std::pair<int, int>> primMesh;  // (mesh, primitive) -> (loaded mesh, material)
    const Json* meshes = doc.find("meshes");
    for (size_t mi = 0; meshes && mi < meshes->size(); mi++) {
        const Json& mesh = meshes->at(mi);
        const Json& prims = mesh.at("primitives");
        for (size_t pi = 0; pi < prims.size(); pi++) {
            const Json& prim = prims.at(pi);
            if (static_cast<int>(prim.number("mode", 4)) != 4) continue;
            const Json& attrs = prim.at("attributes");
            const Accessor pos = accessor(static_cast<size_t>(attrs.at("POSITION").num));
            const bool hasN = attrs.find("NORMAL") != nullptr, hasUV = attrs.find("TEXCOORD_0") != nullptr;
            Accessor nrm, uv, idx;
            if (hasN) nrm = accessor(static_cast<size_t>(attrs.at("NORMAL").num));
            if (hasUV) uv = accessor(static_cast<size_t>(attrs.at("TEXCOORD_0").num));
            const bool indexed = prim.find("indices") != nullptr;
            if (indexed) idx = accessor(static_cast<size_t>(prim.at("indices").num));
            const size_t corners = indexed ? idx.count : pos.count;
            std::vector<Triangle> tris;
            tris.reserve(corners / 3);
            for (size_t t = 0; t + 2 < corners; t += 3) {
                size_t v[3];
                for (int k = 0; k < 3; k++) {
                    v[k] = indexed ? static_cast<size_t>(idx.get(t + k, 0)) : t + k;
                    if (v[k] >= pos.count) fail("glb: vertex index out of range");
                }
                float3 p[3], n[3];
                float2 tc[3];
                for (int k = 0; k < 3; k++) {
                    p[k] = make_float3(static_cast<float>(pos.get(v[k], 0)), static_cast<float>(pos.get(v[k], 1)), static_cast<float>(pos.get(v[k], 2)));
                    n[k] = hasN ? make_float3(static_cast<float>(nrm.get(v[k], 0)), static_cast<float>(nrm.get(v[k], 1)), static_cast<float>(nrm.get(v[k], 2))) : make_float3(0.0f);
                    // aiProcess_FlipUVs, OBJLoader.cpp:219-220
                    tc[k] = hasUV ? make_float2(static_cast<float>(uv.get(v[k], 0)), 1.0f - static_cast<float>(uv.get(v[k], 1))) : make_float2(0.0f, 0.0f);
                }
                tris.emplace_back(p[0], p[1], p[2], n[0], n[1], n[2], tc[0], tc[1], tc[2]);
            }
            primMesh[{mi, pi}] = {static_cast<int>(out.meshes.size()), static_cast<int>(prim.number("material", 0))};
            out.meshes.push_back(std::move(tris));
            const Json* name = mesh.find("name");
            out.meshNames.push_back(name ? name->str : "mesh" + std::to_string(mi));
        }
    }

    const Json& nodes = doc.at("nodes");
    struct Walker {
        const Json& nodes;
        const std::map<std::pair<size_t, size_t>, std::pair<int, int>>& primMesh;
        LoadedScene& out;
        int depth = 0;
        void walk(size_t ni, const M4& parent)
        {
            if (++depth > 256) fail("glb: node hierarchy too deep (cycle?)");
            const Json& node = nodes.at(ni);
            const M4 m = parent * node_local_matrix(node);
            if (const Json* meshRef = node.find("mesh")) {
                LoadedInstance inst;
                decompose_trs(m, inst.position, inst.rotation, inst.scale);
                if (const Json* name = node.find("name")) inst.name = name->str;
                const size_t mi = static_cast<size_t>(meshRef->num);
                for (const auto& kv : primMesh)
                    if (kv.first.first == mi) {
                        inst.mesh = kv.second.first;
                        inst.material = kv.second.second;
                        out.instances.push_back(inst);
                    }
            }
            if (const Json* ch = node.find("children"))
                for (size_t k = 0; k < ch->size(); k++) walk(static_cast<size_t>(ch->at(k).num), m);
            depth--;
        }
    } walker{nodes, primMesh, out};
    const Json& scenes = doc.at("scenes");
    const Json& scene = scenes.at(static_cast<size_t>(doc.number("scene", 0)));
    const Json& roots = scene.at("nodes");
    for (size_t k = 0; k < roots.size(); k++) walker.walk(static_cast<size_t>(roots.at(k).num), M4::identity());
    for (const LoadedInstance& inst : out.instances)
        if (inst.material < 0 || static_cast<size_t>(inst.material) >= out.materials.size()) fail("glb: primitive refers to a material that does not exist");
    return out;
}

// Triangulated / polygonal Wavefront .obj (v / vn / vt / f, fan triangulation), one mesh, Assimp's default material
LoadedScene parse_obj(const std::string& file)
{
    std::ifstream f(file);
    if (!f) fail("cannot open " + file);
    std::vector<float3> v, vn;
    std::vector<float2> vt;
    struct Corner { long vi, ti, ni; };
    std::vector<Corner> faces;  // 3 per triangle
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v" || tag == "vn") {
            float3 p;
            if (!(ss >> p.x >> p.y >> p.z)) fail("obj: malformed " + tag + " line");
            (tag == "v" ? v : vn).push_back(p);
        } else if (tag == "vt") {
            float2 t;
            if (!(ss >> t.x >> t.y)) fail("obj: malformed vt line");
            vt.push_back(t);
        } else if (tag == "f") {
            std::vector<Corner> corners;
            std::string tok;
            while (ss >> tok) {
                Corner c{0, 0, 0};
                const size_t s1 = tok.find('/');
                c.vi = std::strtol(tok.substr(0, s1).c_str(), nullptr, 10);
                if (s1 != std::string::npos) {
                    const size_t s2 = tok.find('/', s1 + 1);
                    const std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) c.ti = std::strtol(t.c_str(), nullptr, 10);
                    if (s2 != std::string::npos) {
                        const std::string n = tok.substr(s2 + 1);
                        if (!n.empty()) c.ni = std::strtol(n.c_str(), nullptr, 10);
                    }
                }
                corners.push_back(c);
            }
            for (size_t k = 1; k + 1 < corners.size(); k++) {
                faces.push_back(corners[0]);
                faces.push_back(corners[k]);
                faces.push_back(corners[k + 1]);
            }
        }
    }
    auto fix = [](long i, size_t n) -> size_t {
        const long r = i > 0 ? i - 1 : static_cast<long>(n) + i;
        if (r < 0 || static_cast<size_t>(r) >= n) fail("obj: index out of range");
        return static_cast<size_t>(r);
    };
    bool allN = !vn.empty(), allT = !vt.empty();
    for (const Corner& c : faces) {
        if (!c.ni) allN = false;
        if (!c.ti) allT = false;
    }
    std::vector<Triangle> tris;
    tris.reserve(faces.size() / 3);
    for (size_t t = 0; t + 2 < faces.size(); t += 3) {
        float3 p[3], n[3];
        float2 tc[3];
        for (int k = 0; k < 3; k++) p[k] = v[fix(faces[t + k].vi, v.size())];
        const float3 fn = allN ? make_float3(0.0f) : face_normal(p[0], p[1], p[2]);
        for (int k = 0; k < 3; k++) {
            n[k] = allN ? vn[fix(faces[t + k].ni, vn.size())] : fn;
            tc[k] = make_float2(0.0f, 0.0f);
            if (allT) {
                const float2 uv = vt[fix(faces[t + k].ti, vt.size())];
                tc[k] = make_float2(uv.x, 1.0f - uv.y);
            }
        }
        tris.emplace_back(p[0], p[1], p[2], n[0], n[1], n[2], tc[0], tc[1], tc[2]);
    }
    if (tris.empty()) fail("obj: " + file + " holds no faces");
    LoadedScene out;
    out.meshes.push_back(std::move(tris));
    out.meshNames.push_back(file);
    Material m;
    m.type = NX_MAT_PLASTIC;
    const double grey[3] = {0.6, 0.6, 0.6};
    set_albedo_like(m, grey, 1.0 - std::sqrt(20.0) / 31.62278, 1.45);
    m.intensity = 0.0f;
    out.materials.push_back(m);
    out.materialDiffuseTexture.push_back(-1);
    out.materialEmissiveTexture.push_back(-1);
    LoadedInstance inst;
    inst.name = file;
    out.instances.push_back(inst);
    return out;
}

bool ends_with(const std::string& s, const char* suffix)
{
    const size_t n = std::strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++)
        if (std::tolower(static_cast<unsigned char>(s[s.size() - n + i])) != suffix[i]) return false;
    return true;
}

}  // namespace

LoadedScene OBJLoader::Parse(const std::string& file)
{
    if (ends_with(file, ".glb")) return parse_glb(file);
    if (ends_with(file, ".obj")) return parse_obj(file);
    fail("unsupported file type (only .glb and .obj): " + file);
}

void OBJLoader::LoadOBJ(const std::string& path, const std::string& filename, Scene* scene, AssetManager* assetManager)
{
    const LoadedScene ls = Parse(path + filename);
    const int materialBase = static_cast<int>(assetManager->GetMaterials().size());
    // textures first: a material carries the id its map got in the manager's diffuse / emissive list (OBJLoader.cpp:134-135, 157-158)
    std::vector<int> textureIds(ls.textures.size(), -1);
    for (size_t i = 0; i < ls.textures.size(); i++) textureIds[i] = assetManager->AddTexture(ls.textures[i]);
    for (size_t i = 0; i < ls.materials.size(); i++) {
        Material m = ls.materials[i];
        if (ls.materialDiffuseTexture[i] >= 0) m.diffuseMapId = textureIds[static_cast<size_t>(ls.materialDiffuseTexture[i])];
        if (ls.materialEmissiveTexture[i] >= 0) m.emissiveMapId = textureIds[static_cast<size_t>(ls.materialEmissiveTexture[i])];
        assetManager->AddMaterial(m);
    }
    // one BVH8 per mesh of the file (OBJLoader.cpp:213-239) — created together, so that a device builder can build them in one go
    std::vector<int32_t> meshIds;
    const int32_t firstBvh = assetManager->CreateBVHs(ls.meshes);
    for (size_t i = 0; i < ls.meshes.size(); i++) meshIds.push_back(assetManager->AddMesh(Mesh(ls.meshNames[i], firstBvh + static_cast<int32_t>(i), -1)));
    for (const LoadedInstance& inst : ls.instances) {
        MeshInstance& mi = scene->CreateMeshInstance(static_cast<uint32_t>(meshIds[static_cast<size_t>(inst.mesh)]));
        mi.name = inst.name.empty() ? ls.meshNames[static_cast<size_t>(inst.mesh)] : inst.name;
        mi.AssignMaterial(materialBase + inst.material);
        mi.SetTransform(inst.position, inst.rotation, inst.scale);
    }
}

}  // namespace nexus
