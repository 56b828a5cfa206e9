// MatrixMarket coordinate I/O: the on-disk format on either side of the path -- the reference's
// NEP data sets are `data/*.mtx` (coordinate real/complex general, written by test/gen_butterfly.jl:6-41 and read with
// MatrixMarket.jl in test/butterfly.jl:57-59, test/polynomial.jl:5-11).  Host only; values are always returned as
// complex128 CSC with 0-based int64 indices, sorted rows, duplicates summed -- the layout feasthip_set_csc takes.
#include <algorithm>
#include <cctype>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/feasthip.h"

namespace {

struct MMFile {
    int64_t rows = 0, cols = 0, nnz = 0;       // nnz after symmetry expansion and duplicate merging
    int field = 0;                              // 0 real, 1 complex, 2 integer, 3 pattern
    std::vector<int64_t> colptr, rowval;
    std::vector<std::complex<double>> val;
};

std::string lower(std::string s) { for (char& c : s) c = (char)std::tolower((unsigned char)c); return s; }

int parse(const char* path, MMFile& M, std::string& err) {
    std::ifstream in(path);
    if (!in) { err = std::string("cannot open ") + path; return FEASTHIP_EARG; }
    std::string line;
    if (!std::getline(in, line)) { err = "empty file"; return FEASTHIP_EARG; }
    std::istringstream hs(line);
    std::string banner, object, format, field, symmetry;
    hs >> banner >> object >> format >> field >> symmetry;
    if (lower(banner) != "%%matrixmarket" || lower(object) != "matrix" || lower(format) != "coordinate") {
        err = "only '%%MatrixMarket matrix coordinate ...' files are supported";
        return FEASTHIP_EARG;
    }
    field = lower(field); symmetry = lower(symmetry);
    M.field = field == "real" ? 0 : field == "complex" ? 1 : field == "integer" ? 2 : field == "pattern" ? 3 : -1;
    int sym = symmetry == "general" ? 0 : symmetry == "symmetric" ? 1 : symmetry == "hermitian" ? 2 : symmetry == "skew-symmetric" ? 3 : -1;
    if (M.field < 0 || sym < 0) { err = "unknown field or symmetry '" + field + " " + symmetry + "'"; return FEASTHIP_EARG; }
    while (std::getline(in, line)) {            // comments, then the size line
        size_t p = line.find_first_not_of(" \t\r");
        if (p == std::string::npos || line[p] == '%') continue;
        break;
    }
    int64_t entries = 0;
    { std::istringstream ss(line); if (!(ss >> M.rows >> M.cols >> entries) || M.rows <= 0 || M.cols <= 0 || entries < 0) { err = "bad size line"; return FEASTHIP_EARG; } }
    struct Ent { int64_t i, j; std::complex<double> v; };
    std::vector<Ent> ents;
    ents.reserve((size_t)entries * (sym ? 2 : 1));
    for (int64_t e = 0; e < entries; ++e) {
        int64_t i, j;
        double re = 1.0, im = 0.0;
        if (!(in >> i >> j)) { err = "unexpected end of file"; return FEASTHIP_EARG; }
        if (M.field == 1) { if (!(in >> re >> im)) { err = "bad complex entry"; return FEASTHIP_EARG; } }
        else if (M.field != 3) { if (!(in >> re)) { err = "bad entry"; return FEASTHIP_EARG; } }
        if (i < 1 || i > M.rows || j < 1 || j > M.cols) { err = "index out of range"; return FEASTHIP_EARG; }
        std::complex<double> v(re, im);
        ents.push_back({i - 1, j - 1, v});
        if (sym && i != j) ents.push_back({j - 1, i - 1, sym == 1 ? v : sym == 2 ? std::conj(v) : -v});
    }
    std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.j != b.j ? a.j < b.j : a.i < b.i; });
    M.colptr.assign((size_t)M.cols + 1, 0);
    for (size_t e = 0; e < ents.size(); ++e) {
        if (!M.rowval.empty() && e > 0 && ents[e].i == ents[e - 1].i && ents[e].j == ents[e - 1].j) { M.val.back() += ents[e].v; continue; }
        M.rowval.push_back(ents[e].i);
        M.val.push_back(ents[e].v);
        M.colptr[(size_t)ents[e].j + 1]++;
    }
    for (int64_t j = 0; j < M.cols; ++j) M.colptr[(size_t)j + 1] += M.colptr[(size_t)j];
    M.nnz = (int64_t)M.rowval.size();
    return 0;
}

thread_local std::string g_mm_error;

}  // namespace

extern "C" {

const char* feasthip_mm_last_error(void) { return g_mm_error.c_str(); }

int feasthip_mm_info(const char* path, int64_t* rows, int64_t* cols, int64_t* nnz, int* field) {
    if (!path) return FEASTHIP_EARG;
    try {
        MMFile M;
        int r = parse(path, M, g_mm_error);
        if (r) return r;
        if (rows) *rows = M.rows;
        if (cols) *cols = M.cols;
        if (nnz) *nnz = M.nnz;
        if (field) *field = M.field;
        return 0;
    } catch (...) { g_mm_error = "out of memory"; return FEASTHIP_EARG; }
}

int feasthip_mm_read_csc(const char* path, int64_t* colptr, int64_t* rowval, void* nzval, int64_t nnz_capacity) {
    if (!path || !colptr || !rowval || !nzval) return FEASTHIP_EARG;
    try {
        MMFile M;
        int r = parse(path, M, g_mm_error);
        if (r) return r;
        if (M.nnz > nnz_capacity) { g_mm_error = "nnz_capacity too small (call feasthip_mm_info first)"; return FEASTHIP_EARG; }
        std::copy(M.colptr.begin(), M.colptr.end(), colptr);
        std::copy(M.rowval.begin(), M.rowval.end(), rowval);
        std::memcpy(nzval, M.val.data(), (size_t)M.nnz * sizeof(std::complex<double>));
        return 0;
    } catch (...) { g_mm_error = "out of memory"; return FEASTHIP_EARG; }
}

int feasthip_mm_write_csc(const char* path, int64_t rows, int64_t cols, const int64_t* colptr, const int64_t* rowval,
                          const void* nzval, int type, int index_base) {
    if (!path || !colptr || !rowval || !nzval || rows <= 0 || cols <= 0) return FEASTHIP_EARG;
    if (type != FEASTHIP_F64 && type != FEASTHIP_C128 && type != FEASTHIP_I64) { g_mm_error = "value type must be F64, C128 or I64"; return FEASTHIP_EARG; }
    FILE* f = std::fopen(path, "w");
    if (!f) { g_mm_error = std::string("cannot open ") + path; return FEASTHIP_EARG; }
    const char* field = type == FEASTHIP_C128 ? "complex" : type == FEASTHIP_I64 ? "integer" : "real";
    std::fprintf(f, "%%%%MatrixMarket matrix coordinate %s general\n", field);          // test/gen_butterfly.jl:15
    int64_t nnz = colptr[cols] - index_base;
    std::fprintf(f, "%lld %lld %lld\n", (long long)rows, (long long)cols, (long long)nnz);
    for (int64_t j = 0; j < cols; ++j)
        for (int64_t e = colptr[j] - index_base; e < colptr[j + 1] - index_base; ++e) {
            long long i = (long long)(rowval[e] - index_base + 1);
            if (type == FEASTHIP_C128) {
                const double* v = static_cast<const double*>(nzval) + 2 * e;
                std::fprintf(f, "%lld %lld %.17g %.17g\n", i, (long long)(j + 1), v[0], v[1]);
            } else if (type == FEASTHIP_I64) {
                std::fprintf(f, "%lld %lld %lld\n", i, (long long)(j + 1), (long long)static_cast<const int64_t*>(nzval)[e]);
            } else {
                std::fprintf(f, "%lld %lld %.17g\n", i, (long long)(j + 1), static_cast<const double*>(nzval)[e]);
            }
        }
    std::fclose(f);
    return 0;
}

}  // extern "C"
