// graph.cc -- Graph container + mtx/bin ingest with the semantics of include/csr_graph.h
// (see gardenia_host.hpp for the line references).
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "gardenia_host.hpp"

VertexId VertexSet::get_intersect_num(const VertexSet &other) const {  // csr_graph.h:24-36
  VertexId num = 0, il = 0, ir = 0;
  while (il < size_ && ir < other.size_) {
    VertexId l = ptr[il], r = other.ptr[ir];
    if (l <= r) il++;
    if (r <= l) ir++;
    if (l == r) num++;
  }
  return num;
}

// .mtx text -> edge arrays, one piece of the file per thread
VertexId read_mtx_edges(const std::string &fname, std::vector<VertexId> &src, std::vector<VertexId> &dst) {
  std::ifstream in(fname.c_str(), std::ios::binary | std::ios::ate);
  if (!in) throw std::runtime_error("File not available: " + fname);
  const size_t size = (size_t)in.tellg();
  std::string buf(size, '\0');
  in.seekg(0);
  in.read(&buf[0], (std::streamsize)size);
  size_t pos = 0;
  auto line_end = [&](size_t p) {
    while (p < size && buf[p] != '\n') p++;
    return p;
  };
  // header: skip '%' lines, then "m n nnz" (csr_graph.h:81-93)
  while (pos < size && buf[pos] == '%') pos = line_end(pos) + 1;
  int m = 0, n = 0;
  long nnz = 0;
  {
    const size_t e = line_end(pos);
    sscanf(buf.substr(pos, e - pos).c_str(), "%d %d %ld", &m, &n, &nnz);
    pos = e + 1;
  }
  if (m != n) printf("Warning, m(%d) != n(%d)\n", m, n);
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  if (size - std::min(pos, size) < ((size_t)1 << 20)) nthreads = 1;
  std::vector<size_t> cut((size_t)nthreads + 1);
  for (int t = 0; t <= nthreads; t++) {
    size_t p = pos + (size > pos ? (size - pos) : 0) / (size_t)nthreads * (size_t)t;
    if (t == nthreads) p = size;
    else if (t > 0 && p > pos) p = line_end(p - 1) + 1;  // move to the start of the next line
    cut[(size_t)t] = std::min(p, size);
  }
  std::vector<std::vector<VertexId> > ps((size_t)nthreads), pd((size_t)nthreads);
  std::vector<char> stopped((size_t)nthreads, 0);
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
  for (int t = 0; t < nthreads; t++) {
    size_t p = cut[(size_t)t];
    const size_t end = cut[(size_t)t + 1];
    std::vector<VertexId> &s = ps[(size_t)t], &d = pd[(size_t)t];
    while (p < end) {
      const size_t e = line_end(p);
      size_t q = p;
      p = e + 1;
      if (e == q || buf[q] == '#') continue;  // csr_graph.h:66-69: blank lines and '#' comments are skipped
      long v[2];
      int got = 0;
      while (got < 2 && q < e) {
        while (q < e && (buf[q] == ' ' || buf[q] == '\t' || buf[q] == '\r')) q++;
        if (q >= e) break;
        bool neg = false;
        if (buf[q] == '-' || buf[q] == '+') neg = buf[q++] == '-';
        if (q >= e || buf[q] < '0' || buf[q] > '9') break;
        long x = 0;
        while (q < e && buf[q] >= '0' && buf[q] <= '9') x = x * 10 + (buf[q++] - '0');
        v[got++] = neg ? -x : x;
      }
      if (got < 2) {  // a line that does not parse ends the reference's read loop (next_line(), :66-72)
        stopped[(size_t)t] = 1;
        break;
      }
      if (v[0] == v[1]) continue;  // self loop, :108
      s.push_back((VertexId)(v[0] - 1));
      d.push_back((VertexId)(v[1] - 1));
    }
  }
  src.clear();
  dst.clear();
  for (int t = 0; t < nthreads; t++) {
    src.insert(src.end(), ps[(size_t)t].begin(), ps[(size_t)t].end());
    dst.insert(dst.end(), pd[(size_t)t].begin(), pd[(size_t)t].end());
    if (stopped[(size_t)t]) break;  // everything behind the first unparsable line is ignored
  }
  return (VertexId)m;
}

// csr_graph.h:122-169 fill_data on the host: rows ascending, duplicates dropped (one global sort of packed
// (src,dst) keys instead of per-row std::sort + the O(deg^2) erase loop)
void Graph::from_edges(VertexId m, std::vector<VertexId> &src, std::vector<VertexId> &dst, bool symmetrize) {
  std::vector<uint64_t> keys;
  keys.reserve(src.size() * (symmetrize ? 2 : 1));
  for (size_t i = 0; i < src.size(); i++) {
    if (src[i] < 0 || src[i] >= m || dst[i] < 0 || dst[i] >= m) throw std::runtime_error("vertex id outside [1, m] in the edge list");
    keys.push_back(((uint64_t)(uint32_t)src[i] << 32) | (uint32_t)dst[i]);
    if (symmetrize) keys.push_back(((uint64_t)(uint32_t)dst[i] << 32) | (uint32_t)src[i]);
  }
  std::sort(keys.begin(), keys.end());
  size_t removed = keys.size();
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  removed -= keys.size();
  printf("Removing redundent edges... %zu redundent edges are removed\n", removed);
  n_vertices = m;
  n_edges = keys.size();
  std::cout << "|V| " << n_vertices << " |E| " << n_edges << "\n";
  vertices.assign((size_t)m + 1, 0);
  edges.resize(n_edges);
  for (size_t i = 0; i < keys.size(); i++) {
    vertices[(size_t)(keys[i] >> 32) + 1]++;
    edges[i] = (VertexId)(uint32_t)keys[i];
  }
  max_degree = 0;
  for (VertexId i = 0; i < m; i++) {
    max_degree = std::max<VertexId>(max_degree, (VertexId)vertices[i + 1]);
    vertices[i + 1] += vertices[i];
  }
}

// the same clean-up through the device builder gdn_graph_from_edges (one radix sort on the GPU); used by
// mtx2bin and by `Graph(..., device_ingest = true)`.  Needs a HIP device.
void Graph::from_edges_device(VertexId m, std::vector<VertexId> &src, std::vector<VertexId> &dst, bool symmetrize) {
  const uint64_t n_in = (uint64_t)src.size() * (symmetrize ? 2 : 1);
  gdn_graph *g = nullptr;
  if (gdn_graph_from_edges(m, src.size(), src.data(), dst.data(), symmetrize ? 1 : 0, &g) != GDN_OK)
    throw std::runtime_error(std::string("gdn_graph_from_edges: ") + gdn_last_error());
  int32_t mm = 0;
  uint64_t nnz = 0;
  gdn_graph_info(g, &mm, &nnz, nullptr, nullptr);
  n_vertices = m;
  n_edges = nnz;
  vertices.assign((size_t)m + 1, 0);
  edges.resize(n_edges);
  const int rc = gdn_graph_download(g, vertices.data(), edges.data());
  gdn_graph_free(g);
  if (rc != GDN_OK) throw std::runtime_error(std::string("gdn_graph_download: ") + gdn_last_error());
  printf("Removing redundent edges... %llu redundent edges are removed\n", (unsigned long long)(n_in - nnz));
  std::cout << "|V| " << n_vertices << " |E| " << n_edges << "\n";
  max_degree = 0;
  for (VertexId i = 0; i < m; i++) max_degree = std::max<VertexId>(max_degree, (VertexId)(vertices[i + 1] - vertices[i]));
}

void Graph::build_reverse_graph() {  // csr_graph.h:170-194
  reverse_vertices.assign((size_t)n_vertices + 1, 0);
  reverse_edges.resize(n_edges);
  for (uint64_t e = 0; e < n_edges; e++) reverse_vertices[edges[e] + 1]++;
  for (VertexId i = 0; i < n_vertices; i++) reverse_vertices[i + 1] += reverse_vertices[i];
  std::vector<uint64_t> cur(reverse_vertices.begin(), reverse_vertices.end() - 1);
  for (VertexId v = 0; v < n_vertices; v++)
    for (uint64_t e = vertices[v]; e < vertices[v + 1]; e++) reverse_edges[cur[edges[e]]++] = v;  // ascending v
}

Graph::Graph(std::string prefix, std::string filetype, bool symmetrize, bool need_reverse, bool device_ingest) {
  if (filetype == "mtx") {
    std::string fname = prefix + ".mtx";
    std::cout << "Reading (.mtx) input file " << fname << "\n";
    std::vector<VertexId> src, dst;
    const VertexId m = read_mtx_edges(fname, src, dst);
    if (device_ingest) from_edges_device(m, src, dst, symmetrize);
    else from_edges(m, src, dst, symmetrize);
  } else if (filetype == "bin") {
    std::ifstream meta((prefix + ".meta.txt").c_str());
    if (!meta) throw std::runtime_error("cannot open " + prefix + ".meta.txt");
    int vid_size = 0;
    meta >> n_vertices >> n_edges >> vid_size >> max_degree;
    std::cout << "|V| " << n_vertices << " |E| " << n_edges << "\n";
    if (vid_size != (int)sizeof(VertexId)) throw std::runtime_error("vertex id size must be 4");
    vertices.resize((size_t)n_vertices + 1);
    edges.resize(n_edges);
    std::ifstream fv((prefix + ".vertex.bin").c_str(), std::ios::binary), fe((prefix + ".edge.bin").c_str(), std::ios::binary);
    if (!fv || !fe) throw std::runtime_error("cannot open " + prefix + ".vertex.bin/.edge.bin");
    fv.read((char *)vertices.data(), sizeof(uint64_t) * vertices.size());
    fe.read((char *)edges.data(), sizeof(VertexId) * edges.size());
  } else {
    throw std::runtime_error("filetype must be mtx or bin");
  }
  if (!symmetrize && need_reverse) {  // csr_graph.h:236-240
    build_reverse_graph();
    directed = true;
    has_reverse = true;
    printf("This graph maintains both incomming and outgoing edge-list\n");
  }
  if (symmetrize) {  // :241-246
    printf("This graph is symmetrized\n");
    reverse_is_alias = true;
    has_reverse = true;
  }
  if (!has_reverse) {  // never hand out garbage for in_rowptr()
    build_reverse_graph();
  }
}

VertexSet Graph::out_neigh(VertexId v, VertexId start_offset) const {  // csr_graph.h:275-281
  uint64_t b = vertices[v], e = vertices[v + 1];
  b += std::min<uint64_t>((uint64_t)start_offset, e - b);
  return VertexSet(edges.data() + b, (VertexId)(e - b));
}

VertexSet Graph::in_neigh(VertexId v) const {
  const std::vector<uint64_t> &rv = reverse_is_alias ? vertices : reverse_vertices;
  const std::vector<VertexId> &re = reverse_is_alias ? edges : reverse_edges;
  return VertexSet(re.data() + rv[v], (VertexId)(rv[v + 1] - rv[v]));
}

void Graph::orientation() {  // src/common/graph.cc:67-113
  std::vector<uint64_t> nv((size_t)n_vertices + 1, 0);
  std::vector<VertexId> ne;
  ne.reserve(n_edges / 2);
  for (VertexId s = 0; s < n_vertices; s++) {
    for (VertexId d : N(s))
      if (get_degree(d) > get_degree(s) || (get_degree(d) == get_degree(s) && d > s)) ne.push_back(d);
    nv[s + 1] = ne.size();
  }
  vertices.swap(nv);
  edges.swap(ne);
  n_edges = edges.size();
  std::cout << "|V| " << n_vertices << " |E| " << n_edges << "\n";
}

void Graph::write_bin(const std::string &prefix) const {
  std::ofstream meta((prefix + ".meta.txt").c_str());
  meta << n_vertices << "\n" << n_edges << "\n" << sizeof(VertexId) << "\n" << max_degree << "\n";
  std::ofstream fv((prefix + ".vertex.bin").c_str(), std::ios::binary), fe((prefix + ".edge.bin").c_str(), std::ios::binary);
  fv.write((const char *)vertices.data(), sizeof(uint64_t) * vertices.size());
  fe.write((const char *)edges.data(), sizeof(VertexId) * edges.size());
}
