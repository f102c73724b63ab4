// Scene graph construction (see scene_graph.hpp).  Field packing per kind:
//   H_SPHERE          f[0..2] center, f[3] radius
//   H_MOVING_SPHERE   f[0..2] center0, f[3..5] center1, f[6] time0, f[7] time1, f[8] radius
//   H_GRAVITY_SPHERE  f[0..2] start, f[3] time0, f[4] radius; table = stored heights
//   H_TRIANGLE        f[0..2] v0, f[3..5] v1, f[6..8] v2, f[9..11] stored unit normal
//   H_XY/XZ/YZ_RECT   f[0] x0, f[1] x1, f[2] y0, f[3] y1, f[4] k   (constructor names)
//   H_RECT_PRISM      f[0..2] p0, f[3..5] p1
//   H_BVH             f[0] time0, f[1] time1, children = the list's objects at from_list time
//   H_TRANSLATE       f[0..2] offset
//   H_ROTATE_Y        f[0] sin_theta, f[1] cos_theta, f[2] angle in degrees
//   H_CONSTANT_MEDIUM f[0] neg_inv_density, mat = the Isotropic phase function
#include "scene_graph.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include "../core/rt_math.hpp"

namespace rtx {

using rt::Vec3;

static GHittable make_h(int32_t kind, int32_t mat) {
  GHittable h;
  h.kind = kind;
  h.mat = mat;
  for (double& x : h.f) x = 0.0;
  return h;
}

// ---------------------------------------------------------------- textures
int32_t SceneGraph::solid_color(const double rgb[3]) {
  GTexture t{};
  t.kind = rt::TEX_SOLID;
  t.a = t.b = -1;
  t.color[0] = rgb[0]; t.color[1] = rgb[1]; t.color[2] = rgb[2];
  textures.push_back(t);
  return (int32_t)textures.size() - 1;
}
int32_t SceneGraph::checker(int32_t even, int32_t odd) {
  if (!valid_texture(even) || !valid_texture(odd)) { error = "checker: bad texture handle"; return -1; }
  GTexture t{};
  t.kind = rt::TEX_CHECKER;
  t.a = even; t.b = odd;
  textures.push_back(t);
  return (int32_t)textures.size() - 1;
}
void perlin_generate(rt::HostRng& rng, rt::FlatPerlin* out) {
  for (int i = 0; i < 256; ++i) {  // perlin.rs:17-19: random_range(-1.0, 1.0), un-normalised
    out->ranvec[i][0] = rt::host_rng_range(rng, -1.0, 1.0);
    out->ranvec[i][1] = rt::host_rng_range(rng, -1.0, 1.0);
    out->ranvec[i][2] = rt::host_rng_range(rng, -1.0, 1.0);
  }
  int32_t* perms[3] = {out->perm_x, out->perm_y, out->perm_z};
  for (int a = 0; a < 3; ++a) {
    int32_t* p = perms[a];
    for (int i = 0; i < 256; ++i) p[i] = i;
    // perlin.rs:79-82: for i in (1..len-1).rev() { target = gen_range(0..i+1); swap }
    for (int i = 254; i >= 1; --i) {
      int target = (int)rt::host_rng_below(rng, (uint64_t)i + 1);
      int32_t tmp = p[i]; p[i] = p[target]; p[target] = tmp;
    }
  }
}
int32_t SceneGraph::noise(double scale) {
  perlins.emplace_back();
  perlin_generate(rng, &perlins.back());
  GTexture t{};
  t.kind = rt::TEX_NOISE;
  t.a = (int32_t)perlins.size() - 1;
  t.b = -1;
  t.scale = scale;
  textures.push_back(t);
  return (int32_t)textures.size() - 1;
}
int32_t SceneGraph::image_from_texels(int32_t w, int32_t h, const double* texels) {
  if (w <= 0 || h <= 0 || !texels) { error = "image: bad dimensions"; return -1; }
  GImage im;
  im.width = w; im.height = h;
  im.texels.assign(texels, texels + (size_t)3 * w * h);
  images.push_back(std::move(im));
  GTexture t{};
  t.kind = rt::TEX_IMAGE;
  t.a = (int32_t)images.size() - 1;
  t.b = -1;
  textures.push_back(t);
  return (int32_t)textures.size() - 1;
}
bool read_ppm_p3(const char* path, int32_t* w, int32_t* h, std::vector<double>* texels, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { *err = std::string("Couldn't open the file: ") + path; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  std::string contents = ss.str();
  // screen.rs:66-72: split("\n"); skip line 0; line 1 = "w h"; skip line 2; rest = numbers.
  size_t p0 = contents.find('\n');
  if (p0 == std::string::npos) { *err = "ppm: truncated header"; return false; }
  size_t p1 = contents.find('\n', p0 + 1);
  if (p1 == std::string::npos) { *err = "ppm: truncated header"; return false; }
  size_t p2 = contents.find('\n', p1 + 1);
  if (p2 == std::string::npos) { *err = "ppm: truncated header"; return false; }
  std::string wh = contents.substr(p0 + 1, p1 - p0 - 1);
  long ww = 0, hh = 0;
  if (sscanf(wh.c_str(), "%ld %ld", &ww, &hh) != 2 || ww <= 0 || hh <= 0) { *err = "ppm: bad size line"; return false; }
  *w = (int32_t)ww; *h = (int32_t)hh;
  texels->clear();
  texels->reserve((size_t)3 * ww * hh);
  const char* s = contents.c_str() + p2 + 1;
  char* end = nullptr;
  for (size_t n = 0; n < (size_t)3 * ww * hh; ++n) {
    double v = strtod(s, &end);
    if (end == s) { *err = "ppm: not enough pixel values"; return false; }
    texels->push_back(v);
    s = end;
  }
  return true;
}
int32_t SceneGraph::image_from_ppm(const char* path) {
  int32_t w, h;
  std::vector<double> tx;
  if (!read_ppm_p3(path, &w, &h, &tx, &error)) return -1;
  return image_from_texels(w, h, tx.data());
}

// ---------------------------------------------------------------- materials
int32_t SceneGraph::lambertian(int32_t tex) {
  if (!valid_texture(tex)) { error = "lambertian: bad texture handle"; return -1; }
  GMaterial m{};
  m.kind = rt::MAT_LAMBERTIAN; m.tex = tex;
  materials.push_back(m);
  return (int32_t)materials.size() - 1;
}
int32_t SceneGraph::metal(const double albedo[3], double fuzz) {
  GMaterial m{};
  m.kind = rt::MAT_METAL; m.tex = -1;
  m.albedo[0] = albedo[0]; m.albedo[1] = albedo[1]; m.albedo[2] = albedo[2];
  m.param = fuzz < 1.0 ? fuzz : 1.0;  // hit.rs:1063
  materials.push_back(m);
  return (int32_t)materials.size() - 1;
}
int32_t SceneGraph::dielectric(double ir) {
  GMaterial m{};
  m.kind = rt::MAT_DIELECTRIC; m.tex = -1; m.param = ir;
  materials.push_back(m);
  return (int32_t)materials.size() - 1;
}
int32_t SceneGraph::diffuse_light(int32_t tex) {
  if (!valid_texture(tex)) { error = "diffuse_light: bad texture handle"; return -1; }
  GMaterial m{};
  m.kind = rt::MAT_DIFFUSE_LIGHT; m.tex = tex;
  materials.push_back(m);
  return (int32_t)materials.size() - 1;
}
int32_t SceneGraph::isotropic(int32_t tex) {
  if (!valid_texture(tex)) { error = "isotropic: bad texture handle"; return -1; }
  GMaterial m{};
  m.kind = rt::MAT_ISOTROPIC; m.tex = tex;
  materials.push_back(m);
  return (int32_t)materials.size() - 1;
}

// ---------------------------------------------------------------- hittables
int32_t SceneGraph::sphere(const double c[3], double radius, int32_t mat) {
  if (!valid_material(mat)) { error = "sphere: bad material handle"; return -1; }
  GHittable h = make_h(H_SPHERE, mat);
  h.f[0] = c[0]; h.f[1] = c[1]; h.f[2] = c[2]; h.f[3] = radius;
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::moving_sphere(const double c0[3], const double c1[3], double t0, double t1,
                                  double radius, int32_t mat) {
  if (!valid_material(mat)) { error = "moving_sphere: bad material handle"; return -1; }
  GHittable h = make_h(H_MOVING_SPHERE, mat);
  for (int i = 0; i < 3; ++i) { h.f[i] = c0[i]; h.f[3 + i] = c1[i]; }
  h.f[6] = t0; h.f[7] = t1; h.f[8] = radius;
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
// GravitySphere::new (hit.rs:340-367): the height table is simulated here exactly as the reference does it.
int32_t SceneGraph::gravity_sphere(const double start[3], double time0, double radius, int32_t mat) {
  if (!valid_material(mat)) { error = "gravity_sphere: bad material handle"; return -1; }
  GHittable h = make_h(H_GRAVITY_SPHERE, mat);
  h.f[0] = start[0]; h.f[1] = start[1]; h.f[2] = start[2]; h.f[3] = time0; h.f[4] = radius;
  h.table.push_back(start[1]);
  const double incr = 0.001;
  double t = time0, y = start[1], vel = 0.0;
  while (t < 100.0) {
    t += incr;
    vel -= 0.000001;
    if (y - 1.0 * radius <= 0.0) vel *= -0.92;
    y = rt::rt_fmax(1.0 * radius, y + vel);
    h.table.push_back(y);
  }
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::triangle(const double v0[3], const double v1[3], const double v2[3], int32_t mat) {
  if (!valid_material(mat)) { error = "triangle: bad material handle"; return -1; }
  GHittable h = make_h(H_TRIANGLE, mat);
  Vec3 a0 = rt::v3(v0[0], v0[1], v0[2]), a1 = rt::v3(v1[0], v1[1], v1[2]), a2 = rt::v3(v2[0], v2[1], v2[2]);
  // hit.rs:97-99: normal = (v1 - v0).cross(v2 - v0).unit()
  Vec3 n = rt::unit(rt::cross(a1 - a0, a2 - a0));
  for (int i = 0; i < 3; ++i) { h.f[i] = v0[i]; h.f[3 + i] = v1[i]; h.f[6 + i] = v2[i]; }
  h.f[9] = n.x; h.f[10] = n.y; h.f[11] = n.z;
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::rect(int32_t kind, double a0, double a1, double b0, double b1, double k, int32_t mat) {
  if (kind != H_XY_RECT && kind != H_XZ_RECT && kind != H_YZ_RECT) { error = "rect: bad kind"; return -1; }
  if (!valid_material(mat)) { error = "rect: bad material handle"; return -1; }
  GHittable h = make_h(kind, mat);
  h.f[0] = a0; h.f[1] = a1; h.f[2] = b0; h.f[3] = b1; h.f[4] = k;
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::rect_prism(const double p0[3], const double p1[3], int32_t mat) {
  if (!valid_material(mat)) { error = "rect_prism: bad material handle"; return -1; }
  GHittable h = make_h(H_RECT_PRISM, mat);
  for (int i = 0; i < 3; ++i) { h.f[i] = p0[i]; h.f[3 + i] = p1[i]; }
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::list_new() {
  hittables.push_back(make_h(H_LIST, -1));
  return (int32_t)hittables.size() - 1;
}
bool SceneGraph::list_add(int32_t list, int32_t obj) {
  if (!valid_hittable(list) || hittables[list].kind != H_LIST) { error = "list_add: not a list"; return false; }
  if (!valid_hittable(obj) || obj == list) { error = "list_add: bad object handle"; return false; }
  hittables[list].children.push_back(obj);
  return true;
}
int32_t SceneGraph::bvh_from_list(int32_t list, double time0, double time1) {
  if (!valid_hittable(list) || hittables[list].kind != H_LIST) { error = "bvh_from_list: not a list"; return -1; }
  if (hittables[list].children.empty()) { error = "bvh_from_list: empty list (the reference panics)"; return -1; }
  GHittable h = make_h(H_BVH, -1);
  h.f[0] = time0; h.f[1] = time1;
  h.children = hittables[list].children;
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::translate(const double offset[3], int32_t obj) {
  if (!valid_hittable(obj)) { error = "translate: bad object handle"; return -1; }
  GHittable h = make_h(H_TRANSLATE, -1);
  h.f[0] = offset[0]; h.f[1] = offset[1]; h.f[2] = offset[2];
  h.children.push_back(obj);
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::rotate_y(double angle_deg, int32_t obj) {
  if (!valid_hittable(obj)) { error = "rotate_y: bad object handle"; return -1; }
  GHittable h = make_h(H_ROTATE_Y, -1);
  double angle = rt::rt_to_radians(angle_deg);  // hit.rs:844-846
  h.f[0] = rt::rt_sin(angle);
  h.f[1] = rt::rt_cos(angle);
  h.f[2] = angle_deg;
  h.children.push_back(obj);
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}
int32_t SceneGraph::constant_medium(const double rgb[3], double density, int32_t boundary) {
  if (!valid_hittable(boundary)) { error = "constant_medium: bad boundary handle"; return -1; }
  int32_t tex = solid_color(rgb);
  int32_t phase = isotropic(tex);  // hit.rs:948
  GHittable h = make_h(H_CONSTANT_MEDIUM, phase);
  h.f[0] = -1.0 / density;  // hit.rs:949
  h.children.push_back(boundary);
  hittables.push_back(std::move(h));
  return (int32_t)hittables.size() - 1;
}

int32_t SceneGraph::triangle_mesh(const double* vertices, int64_t n_vertices, const int64_t* faces,
                                  int64_t n_faces, int32_t mat) {
  if (!valid_material(mat)) { error = "triangle_mesh: bad material handle"; return -1; }
  int32_t list = list_new();
  hittables.reserve(hittables.size() + (size_t)n_faces);
  hittables[list].children.reserve((size_t)n_faces);
  for (int64_t fidx = 0; fidx < n_faces; ++fidx) {
    int64_t a = faces[3 * fidx], b = faces[3 * fidx + 1], c = faces[3 * fidx + 2];
    if (a < 0 || b < 0 || c < 0 || a >= n_vertices || b >= n_vertices || c >= n_vertices) {
      error = "triangle_mesh: face index out of range";
      return -1;
    }
    int32_t t = triangle(vertices + 3 * a, vertices + 3 * b, vertices + 3 * c, mat);
    hittables[list].children.push_back(t);
  }
  return list;
}

// model.rs:13-62: ASCII PLY.  Header lines are split on single spaces; "element vertex N" /
// "element face N" give the counts; each vertex line contributes its first three tokens
// times `scale`; each face line contributes tokens 1..3 (token 0 is the vertex count).
// Beyond the reference: "format binary_little_endian 1.0" files are read too --
// the first three vertex properties are x, y, z (any scalar type), further vertex properties are
// skipped, a face is a list (count, indices...) whose first three indices are used.
namespace {
int ply_type_size(const std::string& t) {
  if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
  if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
  if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
  if (t == "double" || t == "float64") return 8;
  return 0;
}
double ply_read_scalar(const std::string& t, const unsigned char* p) {
  if (t == "float" || t == "float32") { float v; memcpy(&v, p, 4); return (double)v; }
  if (t == "double" || t == "float64") { double v; memcpy(&v, p, 8); return v; }
  if (t == "char" || t == "int8") { int8_t v; memcpy(&v, p, 1); return (double)v; }
  if (t == "uchar" || t == "uint8") { uint8_t v; memcpy(&v, p, 1); return (double)v; }
  if (t == "short" || t == "int16") { int16_t v; memcpy(&v, p, 2); return (double)v; }
  if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, p, 2); return (double)v; }
  if (t == "int" || t == "int32") { int32_t v; memcpy(&v, p, 4); return (double)v; }
  uint32_t v; memcpy(&v, p, 4); return (double)v;
}
}  // namespace

int32_t SceneGraph::triangle_model(const char* path, double scale) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { error = std::string("Couldn't open the file: ") + path; return -1; }
  std::string line;
  int64_t vertex_count = 0, face_count = 0;
  bool got_end = false, binary = false;
  std::vector<std::string> vertex_props;
  std::string face_count_type = "uchar", face_index_type = "int";
  int in_element = 0;  // 1 vertex, 2 face
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line == "end_header") { got_end = true; break; }
    char a[64] = {0}, b[64] = {0}, c[64] = {0}, d[64] = {0}, e[64] = {0};
    int n = sscanf(line.c_str(), "%63s %63s %63s %63s %63s", a, b, c, d, e);
    if (n >= 2 && strcmp(a, "format") == 0) {
      if (strcmp(b, "binary_little_endian") == 0) binary = true;
      else if (strcmp(b, "ascii") != 0) { error = std::string("ply: unsupported format ") + b; return -1; }
    } else if (n >= 3 && strcmp(a, "element") == 0) {
      if (strcmp(b, "vertex") == 0) { vertex_count = atoll(c); in_element = 1; }
      else if (strcmp(b, "face") == 0) { face_count = atoll(c); in_element = 2; }
      else in_element = 0;
    } else if (n >= 3 && strcmp(a, "property") == 0) {
      if (in_element == 1) vertex_props.push_back(b);
      else if (in_element == 2 && strcmp(b, "list") == 0 && n >= 4) { face_count_type = c; face_index_type = d; }
    }
  }
  if (!got_end) { error = "ply: no end_header"; return -1; }
  if (vertex_count < 0 || face_count < 0) { error = "ply: negative element count"; return -1; }
  std::vector<double> verts((size_t)3 * vertex_count);
  std::vector<int64_t> faces((size_t)3 * face_count);
  if (!binary) {
    for (int64_t i = 0; i < vertex_count; ++i) {
      if (!std::getline(f, line)) { error = "ply: truncated vertex list"; return -1; }
      double x, y, z;
      if (sscanf(line.c_str(), "%lf %lf %lf", &x, &y, &z) != 3) { error = "ply: bad vertex line"; return -1; }
      verts[3 * i] = x * scale; verts[3 * i + 1] = y * scale; verts[3 * i + 2] = z * scale;
    }
    for (int64_t i = 0; i < face_count; ++i) {
      if (!std::getline(f, line)) { error = "ply: truncated face list"; return -1; }
      long long n, a, b, c;
      if (sscanf(line.c_str(), "%lld %lld %lld %lld", &n, &a, &b, &c) != 4) { error = "ply: bad face line"; return -1; }
      faces[3 * i] = a; faces[3 * i + 1] = b; faces[3 * i + 2] = c;
    }
  } else {
    if (vertex_props.size() < 3) { error = "ply: a vertex needs at least x y z"; return -1; }
    size_t stride = 0;
    for (const std::string& t : vertex_props) {
      int sz = ply_type_size(t);
      if (sz == 0) { error = "ply: unknown vertex property type " + t; return -1; }
      stride += (size_t)sz;
    }
    std::vector<unsigned char> buf(stride);
    for (int64_t i = 0; i < vertex_count; ++i) {
      if (!f.read((char*)buf.data(), (std::streamsize)stride)) { error = "ply: truncated vertex data"; return -1; }
      size_t off = 0;
      for (int k = 0; k < 3; ++k) {
        verts[3 * i + k] = ply_read_scalar(vertex_props[k], buf.data() + off) * scale;
        off += (size_t)ply_type_size(vertex_props[k]);
      }
    }
    int csz = ply_type_size(face_count_type), isz = ply_type_size(face_index_type);
    if (csz == 0 || isz == 0) { error = "ply: unknown face list types"; return -1; }
    std::vector<unsigned char> fb(8 * 256);
    for (int64_t i = 0; i < face_count; ++i) {
      unsigned char cb[8];
      if (!f.read((char*)cb, csz)) { error = "ply: truncated face data"; return -1; }
      int64_t n = (int64_t)ply_read_scalar(face_count_type, cb);
      if (n < 3 || n > 255) { error = "ply: face with fewer than 3 (or more than 255) vertices"; return -1; }
      if (!f.read((char*)fb.data(), (std::streamsize)(n * isz))) { error = "ply: truncated face data"; return -1; }
      for (int k = 0; k < 3; ++k) faces[3 * i + k] = (int64_t)ply_read_scalar(face_index_type, fb.data() + (size_t)k * isz);
    }
  }
  const double grey[3] = {0.2, 0.2, 0.2};  // model.rs:72
  int32_t mat = lambertian(solid_color(grey));
  return triangle_mesh(verts.data(), vertex_count, faces.data(), face_count, mat);
}

}  // namespace rtx
