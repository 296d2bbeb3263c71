// graph.cc -- Graph container + mtx/bin ingest with the semantics of include/csr_graph.h
// (see gardenia_host.hpp for the line references).
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

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

void Graph::from_edges(VertexId m, std::vector<std::pair<VertexId, VertexId> > &el) {
  // csr_graph.h:122-169 fill_data: rows ascending, duplicates dropped
  std::sort(el.begin(), el.end());
  size_t removed = el.size();
  el.erase(std::unique(el.begin(), el.end()), el.end());
  removed -= el.size();
  printf("Removing redundent edges... %zu redundent edges are removed\n", removed);
  n_vertices = m;
  n_edges = el.size();
  std::cout << "|V| " << n_vertices << " |E| " << n_edges << "\n";
  vertices.assign((size_t)m + 1, 0);
  edges.resize(n_edges);
  for (size_t i = 0; i < el.size(); i++) {
    vertices[el[i].first + 1]++;
    edges[i] = el[i].second;
  }
  max_degree = 0;
  for (VertexId i = 0; i < m; i++) {
    max_degree = std::max<VertexId>(max_degree, (VertexId)vertices[i + 1]);
    vertices[i + 1] += vertices[i];
  }
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

Graph::Graph(std::string prefix, std::string filetype, bool symmetrize, bool need_reverse) {
  if (filetype == "mtx") {
    std::string fname = prefix + ".mtx";
    std::cout << "Reading (.mtx) input file " << fname << "\n";
    std::ifstream in(fname.c_str());
    if (!in) throw std::runtime_error("File not available: " + fname);
    std::string line;
    getline(in, line);
    while (!line.empty() && line[0] == '%') getline(in, line);  // csr_graph.h:81-87
    int m = 0, n = 0;
    long nnz = 0;
    sscanf(line.c_str(), "%d %d %ld", &m, &n, &nnz);
    if (m != n) printf("Warning, m(%d) != n(%d)\n", m, n);
    std::vector<std::pair<VertexId, VertexId> > el;
    el.reserve((size_t)(symmetrize ? 2 * nnz : nnz));
    while (getline(in, line)) {
      if (line.empty() || line[0] == '#') continue;  // csr_graph.h:64-67
      std::istringstream iss(line);
      VertexId a, b;
      if (!(iss >> a >> b)) break;
      if (a == b) continue;  // self loop, :108
      el.push_back(std::make_pair(a - 1, b - 1));
      if (symmetrize) el.push_back(std::make_pair(b - 1, a - 1));
    }
    from_edges(m, el);
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
