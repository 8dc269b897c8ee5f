// ucg_io.cpp -- the on-disk formats of atom style "ucg" (SURVEY.md section 8 row f4), host code.
//
//  * native text dump with the UCG keywords  ucgstate (INT) / ucgl / ucgp (DOUBLE)  and thresholds on them:
//    dump_custom.cpp:1672-1688 (keywords), :1182-1209 + :2150-2155 (thresholds), :3552-3578 (pack),
//    header :650-670, default column formats "%d" / "%g" :142-152, line assembly :1395-1417;
//  * read_dump of those keywords: read_dump.cpp:1344-1349 (names), :823-930 (replace by atom ID, image flags,
//    trim), reader_native.cpp:289-433 (column lookup, x / xs / xu / xsu);
//  * data file sections of the atom style: Atoms "id mol type q x y z ucgstate ucgl ucgml [ix iy iz]",
//    Velocities "id vx vy vz ucgvl" (UCG/atom_vec_ucg.cpp:87-90) and what data_atom_post does to a freshly read
//    atom (clamp ucgl to [0,1] and ucgstate to {0,1}, ucgp = -1: UCG/atom_vec_ucg.cpp:145-170);
//  * restart: the per-atom restart fields ucgstate ucgl ucgml ucgvl ucgp (UCG/atom_vec_ucg.cpp:85) next to the
//    standard ones, in a self-describing binary container (LAMMPS' own restart layout is version-bound).
//
// No device code and no context: every call works on caller-owned host arrays (struct ucg_io_atoms), so the
// LAMMPS glue, the resident loop's host side and the CPU tests use the same entry points.
#include "../../include/ucg_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct IoError {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw IoError{code, buf};
}

int report(const IoError &e, char *err, int errcap) {
    if (err && errcap > 0) snprintf(err, (size_t) errcap, "%s", e.msg.c_str());
    return e.code;
}

std::vector<std::string> split_ws(const std::string &s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char) s[i])) i++;
        if (i >= s.size()) break;
        size_t j = i;
        if (s[i] == '"') {  // quoted word (dump_modify format line "...")
            j = s.find('"', i + 1);
            if (j == std::string::npos) j = s.size();
            out.push_back(s.substr(i + 1, j - i - 1));
            i = j + 1;
            continue;
        }
        while (j < s.size() && !isspace((unsigned char) s[j])) j++;
        out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

struct File {
    FILE *fp = nullptr;
    File(const char *path, const char *mode) {
        fp = fopen(path, mode);
        if (!fp) fail(UCG_ERR_INPUT, "Cannot open file %s", path);
    }
    ~File() {
        if (fp) fclose(fp);
    }
    // writers call this before returning: a full disk shows up in ferror / fclose, not in fprintf's callers
    void finish(const char *what) {
        const bool bad = ferror(fp) != 0;
        const int rc = fclose(fp);
        fp = nullptr;
        if (bad || rc != 0) fail(UCG_ERR_INPUT, "Write error on %s", what);
    }
};

// ------------------------------------------------------------------ dump columns

enum ColKind {
    C_ID, C_MOL, C_TYPE, C_MASS, C_X, C_Y, C_Z, C_XS, C_YS, C_ZS, C_XU, C_YU, C_ZU, C_XSU, C_YSU, C_ZSU,
    C_IX, C_IY, C_IZ, C_VX, C_VY, C_VZ, C_FX, C_FY, C_FZ, C_Q, C_UCGSTATE, C_UCGL, C_UCGP,
    C_UCGVL, C_UCGML, C_UCGFORCE, C_NONE
};

struct ColDef {
    const char *name;
    ColKind kind;
    bool is_int;
};

// the reference's dump keywords for this atom style, plus the three remaining property_atom names
// (UCG/atom_vec_ucg.cpp:172-181) which LAMMPS reaches through compute property/atom
const ColDef COLS[] = {
    {"id", C_ID, true},       {"mol", C_MOL, true},     {"type", C_TYPE, true},   {"mass", C_MASS, false},
    {"x", C_X, false},        {"y", C_Y, false},        {"z", C_Z, false},        {"xs", C_XS, false},
    {"ys", C_YS, false},      {"zs", C_ZS, false},      {"xu", C_XU, false},      {"yu", C_YU, false},
    {"zu", C_ZU, false},      {"xsu", C_XSU, false},    {"ysu", C_YSU, false},    {"zsu", C_ZSU, false},
    {"ix", C_IX, true},       {"iy", C_IY, true},       {"iz", C_IZ, true},       {"vx", C_VX, false},
    {"vy", C_VY, false},      {"vz", C_VZ, false},      {"fx", C_FX, false},      {"fy", C_FY, false},
    {"fz", C_FZ, false},      {"q", C_Q, false},        {"ucgstate", C_UCGSTATE, true},
    {"ucgl", C_UCGL, false},  {"ucgp", C_UCGP, false},  {"ucgvl", C_UCGVL, false}, {"ucgml", C_UCGML, false},
    {"ucgforce", C_UCGFORCE, false},
};

const ColDef *find_col(const std::string &w) {
    for (const ColDef &c : COLS)
        if (w == c.name) return &c;
    return nullptr;
}

// value of one dump attribute of atom i, as the double DumpCustom packs into its buffer
double attr_value(const ucg_io_atoms *a, ColKind k, long long i) {
    auto need = [&](const void *p, const char *what) {
        if (!p) fail(UCG_ERR_INPUT, "Dumping an atom property that isn't allocated (%s)", what);
    };
    auto prd = [&](int d) { return a->boxhi[d] - a->boxlo[d]; };
    auto img = [&](int d) { return a->image ? a->image[3 * i + d] : 0; };
    switch (k) {
        case C_ID: need(a->id, "id"); return a->id[i];
        case C_MOL: need(a->molecule, "mol"); return a->molecule[i];
        case C_TYPE: need(a->type, "type"); return a->type[i];
        case C_MASS: need(a->mass, "mass"); need(a->type, "type"); return a->mass[a->type[i]];
        case C_X: case C_Y: case C_Z: need(a->x, "x"); return a->x[3 * i + (k - C_X)];
        case C_XS: case C_YS: case C_ZS: {
            need(a->x, "x");
            int d = k - C_XS;
            return (a->x[3 * i + d] - a->boxlo[d]) * (1.0 / prd(d));  // pack_xs: (x - boxlo) * invprd
        }
        case C_XU: case C_YU: case C_ZU: {
            need(a->x, "x");
            int d = k - C_XU;
            return a->x[3 * i + d] + img(d) * prd(d);
        }
        case C_XSU: case C_YSU: case C_ZSU: {
            need(a->x, "x");
            int d = k - C_XSU;
            return (a->x[3 * i + d] - a->boxlo[d]) * (1.0 / prd(d)) + img(d);
        }
        case C_IX: case C_IY: case C_IZ: return img(k - C_IX);
        case C_VX: case C_VY: case C_VZ: need(a->v, "v"); return a->v[3 * i + (k - C_VX)];
        case C_FX: case C_FY: case C_FZ: need(a->f, "f"); return a->f[3 * i + (k - C_FX)];
        case C_Q: return a->q ? a->q[i] : 0.0;
        case C_UCGSTATE: need(a->ucgstate, "ucgstate"); return a->ucgstate[i];
        case C_UCGL: need(a->ucgl, "ucgl"); return a->ucgl[i];
        case C_UCGP: need(a->ucgp, "ucgp"); return a->ucgp[i];
        case C_UCGVL: need(a->ucgvl, "ucgvl"); return a->ucgvl[i];
        case C_UCGML: need(a->ucgml, "ucgml"); return a->ucgml[i];
        case C_UCGFORCE: need(a->ucgforce, "ucgforce"); return a->ucgforce[i];
        default: break;
    }
    return 0.0;
}

enum ThreshOp { LT, LE, GT, GE, EQ, NEQ };
struct Thresh {
    ColKind kind;
    ThreshOp op;
    double value;
};

struct DumpSpec {
    std::vector<const ColDef *> cols;
    std::vector<std::string> fmt;  // one printf format per column, trailing blank except the last
    std::vector<Thresh> thresh;
    bool sort_id = false;
    std::string boundary = "pp pp pp";
};

void parse_dump_spec(const char *columns, const char *modify, DumpSpec &S) {
    for (const std::string &w : split_ws(columns ? columns : "")) {
        const ColDef *c = find_col(w);
        if (!c) fail(UCG_ERR_INPUT, "Invalid attribute %s in dump custom command", w.c_str());
        S.cols.push_back(c);
    }
    if (S.cols.empty()) fail(UCG_ERR_INPUT, "No dump custom arguments specified");
    std::string int_user, float_user, line_user;
    std::vector<std::string> col_user(S.cols.size());
    // one dump_modify keyword group per line
    std::string all = modify ? modify : "";
    size_t pos = 0;
    while (pos <= all.size()) {
        size_t nl = all.find('\n', pos);
        if (nl == std::string::npos) nl = all.size();
        std::vector<std::string> w = split_ws(all.substr(pos, nl - pos));
        pos = nl + 1;
        if (w.empty()) continue;
        if (w[0] == "thresh") {
            if (w.size() == 2 && w[1] == "none") {
                S.thresh.clear();
                continue;
            }
            if (w.size() != 4) fail(UCG_ERR_INPUT, "Illegal dump_modify thresh command");
            const ColDef *c = find_col(w[1]);
            if (!c) fail(UCG_ERR_INPUT, "Invalid dump_modify thresh attribute: %s", w[1].c_str());
            Thresh t{c->kind, LT, 0.0};
            if (w[2] == "<") t.op = LT;
            else if (w[2] == "<=") t.op = LE;
            else if (w[2] == ">") t.op = GT;
            else if (w[2] == ">=") t.op = GE;
            else if (w[2] == "==") t.op = EQ;
            else if (w[2] == "!=") t.op = NEQ;
            else fail(UCG_ERR_INPUT, "Invalid dump_modify thresh operator");
            char *end = nullptr;
            t.value = strtod(w[3].c_str(), &end);
            if (end == w[3].c_str() || *end) fail(UCG_ERR_INPUT, "Invalid dump_modify thresh value %s", w[3].c_str());
            S.thresh.push_back(t);
        } else if (w[0] == "sort") {
            if (w.size() != 2) fail(UCG_ERR_INPUT, "Illegal dump_modify sort command");
            if (w[1] == "id") S.sort_id = true;
            else if (w[1] == "off") S.sort_id = false;
            else fail(UCG_ERR_UNSUPPORTED, "dump_modify sort %s is not supported (id | off)", w[1].c_str());
        } else if (w[0] == "format") {
            if (w.size() == 2 && w[1] == "none") {
                int_user.clear(); float_user.clear(); line_user.clear();
                for (auto &s : col_user) s.clear();
                continue;
            }
            if (w.size() != 3) fail(UCG_ERR_INPUT, "Illegal dump_modify format command");
            if (w[1] == "line") line_user = w[2];
            else if (w[1] == "int") int_user = w[2];
            else if (w[1] == "float") float_user = w[2];
            else {
                char *end = nullptr;
                long m = strtol(w[1].c_str(), &end, 10);
                if (*end || m < 1 || m > (long) S.cols.size()) fail(UCG_ERR_INPUT, "Illegal dump_modify format command");
                col_user[m - 1] = w[2];
            }
        } else if (w[0] == "boundary") {  // the header's "pp pp pp" (Domain::boundary_string), e.g. "pp pp ff"
            if (w.size() != 4) fail(UCG_ERR_INPUT, "Illegal boundary string");
            S.boundary = w[1] + " " + w[2] + " " + w[3];
        } else {
            fail(UCG_ERR_UNSUPPORTED, "dump_modify %s is not supported", w[0].c_str());
        }
    }
    // DumpCustom::init_style (dump_custom.cpp:262-290): the line format split into words, each overridden by
    // its column / int / float user format
    std::vector<std::string> words;
    if (!line_user.empty()) {
        words = split_ws(line_user);
        if (words.size() != S.cols.size()) fail(UCG_ERR_INPUT, "Dump_modify format line is too short");
    } else {
        for (const ColDef *c : S.cols) words.push_back(c->is_int ? "%d" : "%g");
    }
    for (size_t i = 0; i < S.cols.size(); i++) {
        std::string f = words[i];
        if (!col_user[i].empty()) f = col_user[i];
        else if (S.cols[i]->is_int && !int_user.empty()) f = int_user;
        else if (!S.cols[i]->is_int && !float_user.empty()) f = float_user;
        if (i + 1 < S.cols.size()) f += " ";
        S.fmt.push_back(f);
    }
}

bool thresh_keep(const Thresh &t, double v) {
    switch (t.op) {
        case LT: return v < t.value;
        case LE: return v <= t.value;
        case GT: return v > t.value;
        case GE: return v >= t.value;
        case EQ: return v == t.value;
        case NEQ: return v != t.value;
    }
    return true;
}

// ------------------------------------------------------------------ text scanning helpers

struct LineReader {
    FILE *fp;
    std::vector<char> buf;
    explicit LineReader(FILE *f) : fp(f), buf(1 << 16) {}
    // next line without its newline, or nullptr at end of file
    char *next() {
        size_t len = 0;
        for (;;) {
            if (!fgets(buf.data() + len, (int) (buf.size() - len), fp)) {
                if (len == 0) return nullptr;
                break;
            }
            len += strlen(buf.data() + len);
            if (len && buf[len - 1] == '\n') break;
            if (len + 1 >= buf.size()) buf.resize(buf.size() * 2);
            else break;  // last line without newline
        }
        while (len && (buf[len - 1] == '\n' || buf[len - 1] == '\r')) buf[--len] = 0;
        return buf.data();
    }
};

bool starts_with(const char *s, const char *prefix) { return strncmp(s, prefix, strlen(prefix)) == 0; }

struct SnapHeader {
    long long timestep = 0, natoms = 0;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    std::string boundary, columns;
    bool triclinic = false;
};

// reads one snapshot header; the stream is left at the first atom line.  false at end of file.
bool read_snap_header(LineReader &R, SnapHeader &H) {
    char *l;
    for (;;) {
        l = R.next();
        if (!l) return false;
        if (starts_with(l, "ITEM: TIMESTEP")) break;
        if (starts_with(l, "ITEM: UNITS") || starts_with(l, "ITEM: TIME")) {
            R.next();
            continue;
        }
        if (*l) fail(UCG_ERR_INPUT, "Dump file is incorrectly formatted: expected ITEM: TIMESTEP, got '%.60s'", l);
    }
    l = R.next();
    if (!l) fail(UCG_ERR_INPUT, "Unexpected end of dump file");
    H.timestep = atoll(l);
    l = R.next();
    if (!l || !starts_with(l, "ITEM: NUMBER OF ATOMS")) fail(UCG_ERR_INPUT, "Dump file is incorrectly formatted");
    l = R.next();
    if (!l) fail(UCG_ERR_INPUT, "Unexpected end of dump file");
    H.natoms = atoll(l);
    l = R.next();
    if (!l || !starts_with(l, "ITEM: BOX BOUNDS")) fail(UCG_ERR_INPUT, "Dump file is incorrectly formatted");
    H.boundary = l + strlen("ITEM: BOX BOUNDS");
    while (!H.boundary.empty() && H.boundary[0] == ' ') H.boundary.erase(0, 1);
    H.triclinic = starts_with(H.boundary.c_str(), "xy xz yz") || starts_with(H.boundary.c_str(), "abc origin");
    if (H.triclinic) fail(UCG_ERR_UNSUPPORTED, "Triclinic dump snapshots are not supported");
    for (int d = 0; d < 3; d++) {
        l = R.next();
        if (!l || sscanf(l, "%lf %lf", &H.lo[d], &H.hi[d]) != 2) fail(UCG_ERR_INPUT, "Dump file is incorrectly formatted");
    }
    l = R.next();
    if (!l || !starts_with(l, "ITEM: ATOMS")) fail(UCG_ERR_INPUT, "Dump file is incorrectly formatted");
    H.columns = l + strlen("ITEM: ATOMS");
    while (!H.columns.empty() && H.columns[0] == ' ') H.columns.erase(0, 1);
    return true;
}

void skip_atoms(LineReader &R, long long n) {
    for (long long i = 0; i < n; i++)
        if (!R.next()) fail(UCG_ERR_INPUT, "Unexpected end of dump file");
}

// positions the stream at the atom lines of the wanted snapshot (timestep < 0: the first one)
void seek_snapshot(LineReader &R, long long timestep, SnapHeader &H, const char *path) {
    for (;;) {
        if (!read_snap_header(R, H)) fail(UCG_ERR_INPUT, "Dump file %s does not contain requested snapshot", path);
        if (timestep < 0 || H.timestep == timestep) return;
        skip_atoms(R, H.natoms);
    }
}

int find_label(const std::vector<std::string> &labels, const char *name) {
    for (size_t i = 0; i < labels.size(); i++)
        if (labels[i] == name) return (int) i;
    return -1;
}

}  // namespace

// ==================================================================================== C ABI

extern "C" {

int ucg_io_dump_write(const char *path, int append, long long timestep, const ucg_io_atoms *a, const char *columns,
                      const char *modify, long long *nwritten, char *err, int errcap) {
    try {
        if (!path || !a) fail(UCG_ERR_INVALID, "ucg_io_dump_write: null argument");
        DumpSpec S;
        parse_dump_spec(columns, modify, S);
        // choose (DumpCustom::count, dump_custom.cpp:741-1320): every threshold must hold
        std::vector<long long> clist;
        clist.reserve((size_t) a->n);
        for (long long i = 0; i < a->n; i++) {
            bool keep = true;
            for (const Thresh &t : S.thresh)
                if (!thresh_keep(t, attr_value(a, t.kind, i))) {
                    keep = false;
                    break;
                }
            if (keep) clist.push_back(i);
        }
        if (S.sort_id) {
            if (!a->id) fail(UCG_ERR_INPUT, "Dump sort id needs atom IDs");
            std::stable_sort(clist.begin(), clist.end(), [&](long long p, long long q) { return a->id[p] < a->id[q]; });
        }
        File F(path, append ? "a" : "w");
        // DumpCustom::header_item (dump_custom.cpp:650-670)
        fprintf(F.fp, "ITEM: TIMESTEP\n%lld\nITEM: NUMBER OF ATOMS\n%lld\n", timestep, (long long) clist.size());
        fprintf(F.fp, "ITEM: BOX BOUNDS %s\n", S.boundary.c_str());
        for (int d = 0; d < 3; d++) fprintf(F.fp, "%1.16e %1.16e\n", a->boxlo[d], a->boxhi[d]);
        std::string cols;
        for (size_t j = 0; j < S.cols.size(); j++) cols += (j ? " " : "") + std::string(S.cols[j]->name);
        fprintf(F.fp, "ITEM: ATOMS %s\n", cols.c_str());
        // DumpCustom::convert_string / write_lines (dump_custom.cpp:1395-1417, :1448-1465)
        std::vector<char> line(256 * S.cols.size() + 16);
        for (long long i : clist) {
            size_t off = 0;
            for (size_t j = 0; j < S.cols.size(); j++) {
                double v = attr_value(a, S.cols[j]->kind, i);
                int w;
                if (S.cols[j]->is_int) w = snprintf(&line[off], line.size() - off, S.fmt[j].c_str(), static_cast<int>(v));
                else w = snprintf(&line[off], line.size() - off, S.fmt[j].c_str(), v);
                if (w < 0 || (size_t) w >= line.size() - off) fail(UCG_ERR_INPUT, "Dump line too long for column format");
                off += (size_t) w;
            }
            line[off++] = '\n';
            fwrite(line.data(), 1, off, F.fp);
        }
        F.finish(path);
        if (nwritten) *nwritten = (long long) clist.size();
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

int ucg_io_dump_scan(const char *path, long long *timesteps, long long *natoms, int cap, int *nfound, char *err,
                     int errcap) {
    try {
        File F(path, "r");
        LineReader R(F.fp);
        SnapHeader H;
        int n = 0;
        while (read_snap_header(R, H)) {
            if (n < cap) {
                if (timesteps) timesteps[n] = H.timestep;
                if (natoms) natoms[n] = H.natoms;
            }
            n++;
            skip_atoms(R, H.natoms);
        }
        if (nfound) *nfound = n;
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

int ucg_io_dump_header(const char *path, long long timestep, long long *found_timestep, long long *natoms,
                       double *boxlo, double *boxhi, char *columns, int columns_cap, char *err, int errcap) {
    try {
        File F(path, "r");
        LineReader R(F.fp);
        SnapHeader H;
        seek_snapshot(R, timestep, H, path);
        if (found_timestep) *found_timestep = H.timestep;
        if (natoms) *natoms = H.natoms;
        for (int d = 0; d < 3; d++) {
            if (boxlo) boxlo[d] = H.lo[d];
            if (boxhi) boxhi[d] = H.hi[d];
        }
        if (columns && columns_cap > 0) snprintf(columns, (size_t) columns_cap, "%s", H.columns.c_str());
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

/* all columns of one snapshot as doubles, values[natoms][ncolumns] in file order */
int ucg_io_dump_load(const char *path, long long timestep, long long natoms, int ncolumns, double *values, char *err,
                     int errcap) {
    try {
        File F(path, "r");
        LineReader R(F.fp);
        SnapHeader H;
        seek_snapshot(R, timestep, H, path);
        int nc = (int) split_ws(H.columns).size();
        if (H.natoms != natoms || nc != ncolumns) fail(UCG_ERR_INVALID, "ucg_io_dump_load: snapshot is %lld x %d", H.natoms, nc);
        for (long long i = 0; i < natoms; i++) {
            char *l = R.next();
            if (!l) fail(UCG_ERR_INPUT, "Unexpected end of dump file");
            char *p = l;
            for (int j = 0; j < nc; j++) {
                char *end;
                values[i * nc + j] = strtod(p, &end);
                if (end == p) fail(UCG_ERR_INPUT, "Dump file is incorrectly formatted: atom line %lld", i + 1);
                p = end;
            }
        }
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

/* read_dump path timestep fields ... [box yes|no] [replace yes|no] [trim yes|no] [scaled yes|no] [wrapped yes|no]
 * stats[0..3] = atoms in snapshot, replaced, trimmed, new atom count a->n */
int ucg_io_read_dump(const char *path, long long timestep, const char *fields, const char *options, ucg_io_atoms *a,
                     long long *stats, char *err, int errcap) {
    try {
        if (!path || !a || !a->id) fail(UCG_ERR_INVALID, "ucg_io_read_dump: atoms with IDs are required");
        bool boxflag = true, replaceflag = true, trimflag = false, scaled = false, wrapped = true;
        {
            std::vector<std::string> w = split_ws(options ? options : "");
            if (w.size() % 2) fail(UCG_ERR_INPUT, "Illegal read_dump command");
            for (size_t i = 0; i < w.size(); i += 2) {
                bool yes;
                if (w[i + 1] == "yes") yes = true;
                else if (w[i + 1] == "no") yes = false;
                else fail(UCG_ERR_INPUT, "Illegal read_dump command: %s %s", w[i].c_str(), w[i + 1].c_str());
                if (w[i] == "box") boxflag = yes;
                else if (w[i] == "replace") replaceflag = yes;
                else if (w[i] == "trim") trimflag = yes;
                else if (w[i] == "scaled") scaled = yes;
                else if (w[i] == "wrapped") wrapped = yes;
                else if ((w[i] == "purge" || w[i] == "add") && !yes) continue;
                else if (w[i] == "purge" || w[i] == "add")
                    // read_dump.cpp:954: "UCG currently does not support this" -- new atoms would carry no UCG fields
                    fail(UCG_ERR_UNSUPPORTED, "read_dump %s yes is not supported for atom style ucg: load the snapshot with "
                                              "ucg_io_dump_load instead", w[i].c_str());
                else fail(UCG_ERR_INPUT, "Illegal read_dump command: %s", w[i].c_str());
            }
        }
        // requested fields (read_dump.cpp:1300-1349): id is always field 0
        std::vector<const ColDef *> want;
        for (const std::string &w : split_ws(fields ? fields : "")) {
            const ColDef *c = find_col(w);
            bool ok = c && (c->kind == C_X || c->kind == C_Y || c->kind == C_Z || (c->kind >= C_IX && c->kind <= C_Q) ||
                            c->kind == C_UCGSTATE || c->kind == C_UCGL || c->kind == C_UCGP || c->kind == C_TYPE);
            if (!ok) fail(UCG_ERR_INPUT, "Illegal read_dump command: unknown field %s", w.c_str());
            want.push_back(c);
        }
        File F(path, "r");
        LineReader R(F.fp);
        SnapHeader H;
        seek_snapshot(R, timestep, H, path);
        std::vector<std::string> labels = split_ws(H.columns);
        int id_col = find_label(labels, "id");
        if (id_col < 0) fail(UCG_ERR_INPUT, "Read_dump field not found in dump file: id");
        // box first (read_dump.cpp header(): box yes resets the simulation box before atoms are processed)
        double lo[3], hi[3];
        for (int d = 0; d < 3; d++) {
            lo[d] = boxflag ? H.lo[d] : a->boxlo[d];
            hi[d] = boxflag ? H.hi[d] : a->boxhi[d];
        }
        // column of each wanted field; x may come as x / xs / xu / xsu (reader_native.cpp:329-398)
        struct Src {
            int col;
            int xmode;  // 0 plain; for coordinates: 1 = x, 2 = xs, 3 = xu, 4 = xsu
        };
        std::vector<Src> src(want.size());
        for (size_t k = 0; k < want.size(); k++) {
            const ColDef *c = want[k];
            src[k] = {-1, 0};
            if (c->kind == C_X || c->kind == C_Y || c->kind == C_Z) {
                const char axis = "xyz"[c->kind - C_X];
                std::string n1(1, axis), n2 = n1 + "s", n3 = n1 + "u", n4 = n1 + "su";
                int i1 = find_label(labels, n1.c_str()), i2 = find_label(labels, n2.c_str());
                int i3 = find_label(labels, n3.c_str()), i4 = find_label(labels, n4.c_str());
                // preference follows the requested representation (scaled / wrapped), then whatever exists
                int pick[4];
                if (!scaled && wrapped) { pick[0] = 1; pick[1] = 2; pick[2] = 3; pick[3] = 4; }
                else if (scaled && wrapped) { pick[0] = 2; pick[1] = 1; pick[2] = 4; pick[3] = 3; }
                else if (!scaled) { pick[0] = 3; pick[1] = 4; pick[2] = 1; pick[3] = 2; }
                else { pick[0] = 4; pick[1] = 3; pick[2] = 2; pick[3] = 1; }
                int idx[5] = {-1, i1, i2, i3, i4};
                for (int p : pick)
                    if (idx[p] >= 0) {
                        src[k] = {idx[p], p};
                        break;
                    }
            } else {
                src[k].col = find_label(labels, c->name);
            }
            if (src[k].col < 0) fail(UCG_ERR_INPUT, "Read_dump field not found in dump file: %s", c->name);
            if ((c->kind == C_UCGSTATE && !a->ucgstate) || (c->kind == C_UCGL && !a->ucgl) || (c->kind == C_UCGP && !a->ucgp))
                fail(UCG_ERR_INPUT, "Read dump of UCG %s property that isn't supported by atom style",
                     c->kind == C_UCGSTATE ? "state" : c->kind == C_UCGL ? "L" : "P");  // read_dump.cpp:1201-1207
            if ((c->kind >= C_VX && c->kind <= C_VZ && !a->v) || (c->kind >= C_FX && c->kind <= C_FZ && !a->f) ||
                (c->kind == C_Q && !a->q) || ((c->kind == C_X || c->kind == C_Y || c->kind == C_Z) && !a->x))
                fail(UCG_ERR_INPUT, "Read dump of atom property that isn't allocated (%s)", c->name);
        }
        std::unordered_map<int, long long> map;
        map.reserve((size_t) a->n * 2);
        for (long long i = 0; i < a->n; i++) map[a->id[i]] = i;
        std::vector<char> updateflag((size_t) a->n, 0);
        long long nreplace = 0;
        int nc = (int) labels.size();
        std::vector<double> row((size_t) nc);
        for (long long s = 0; s < H.natoms; s++) {
            char *l = R.next();
            if (!l) fail(UCG_ERR_INPUT, "Unexpected end of dump file");
            char *p = l;
            for (int j = 0; j < nc; j++) {
                char *end;
                row[j] = strtod(p, &end);
                if (end == p) fail(UCG_ERR_INPUT, "Dump file is incorrectly formatted: atom line %lld", s + 1);
                p = end;
            }
            auto it = map.find((int) row[id_col]);
            if (it == map.end()) continue;
            long long m = it->second;
            updateflag[m] = 1;
            if (!replaceflag) continue;
            nreplace++;
            int box3[3] = {a->image ? a->image[3 * m] : 0, a->image ? a->image[3 * m + 1] : 0, a->image ? a->image[3 * m + 2] : 0};
            for (size_t k = 0; k < want.size(); k++) {
                double v = row[src[k].col];
                switch (want[k]->kind) {
                    case C_X: case C_Y: case C_Z: {
                        int d = want[k]->kind - C_X;
                        // scaled representations -> box coordinates of the (new) box
                        if (src[k].xmode == 2 || src[k].xmode == 4) v = v * (hi[d] - lo[d]) + lo[d];
                        a->x[3 * m + d] = v;
                        break;
                    }
                    case C_VX: case C_VY: case C_VZ: a->v[3 * m + (want[k]->kind - C_VX)] = v; break;
                    case C_FX: case C_FY: case C_FZ: a->f[3 * m + (want[k]->kind - C_FX)] = v; break;
                    case C_Q: a->q[m] = v; break;
                    case C_IX: case C_IY: case C_IZ: box3[want[k]->kind - C_IX] = static_cast<int>(v); break;
                    case C_UCGSTATE: a->ucgstate[m] = static_cast<int>(v); break;
                    case C_UCGL: a->ucgl[m] = v; break;
                    case C_UCGP: a->ucgp[m] = v; break;
                    case C_TYPE: break;  // read_dump.cpp:858: "1 will be skipped if type"
                    default: break;
                }
            }
            if (!wrapped) box3[0] = box3[1] = box3[2] = 0;  // read_dump.cpp:917
            if (a->image)
                for (int d = 0; d < 3; d++) a->image[3 * m + d] = box3[d];
        }
        if (boxflag)
            for (int d = 0; d < 3; d++) {
                a->boxlo[d] = lo[d];
                a->boxhi[d] = hi[d];
            }
        // trim (read_dump.cpp:925-942): the last atom is copied into the hole, exactly in that order
        long long ntrim = 0, nlocal = a->n;
        if (trimflag) {
            auto copy = [&](long long from, long long to) {
                auto c1 = [&](auto *p) { if (p) p[to] = p[from]; };
                auto c3 = [&](auto *p) { if (p) for (int d = 0; d < 3; d++) p[3 * to + d] = p[3 * from + d]; };
                c1(a->id); c1(a->type); c1(a->molecule); c1(a->ucgstate); c1(a->q); c1(a->ucgl); c1(a->ucgvl);
                c1(a->ucgml); c1(a->ucgp); c1(a->ucgforce); c3(a->x); c3(a->v); c3(a->f); c3(a->image);
            };
            long long i = 0;
            while (i < nlocal) {
                if (!updateflag[i]) {
                    copy(nlocal - 1, i);
                    updateflag[i] = updateflag[nlocal - 1];
                    nlocal--;
                    ntrim++;
                } else i++;
            }
            a->n = nlocal;
        }
        if (stats) {
            stats[0] = H.natoms;
            stats[1] = nreplace;
            stats[2] = ntrim;
            stats[3] = a->n;
        }
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

// ---------------------------------------------------------------------------------- data file

int ucg_io_write_data(const char *path, long long timestep, const char *units, const ucg_io_atoms *a, char *err,
                      int errcap) {
    try {
        if (!path || !a || !a->id || !a->type || !a->x) fail(UCG_ERR_INVALID, "ucg_io_write_data: id, type and x are required");
        File F(path, "w");
        fprintf(F.fp, "LAMMPS data file via libucg_hip, timestep = %lld, units = %s\n\n", timestep, units ? units : "lj");
        fprintf(F.fp, "%lld atoms\n%d atom types\n\n", a->n, a->ntypes);
        const char *ax[3] = {"x", "y", "z"};
        for (int d = 0; d < 3; d++) fprintf(F.fp, "%.17g %.17g %slo %shi\n", a->boxlo[d], a->boxhi[d], ax[d], ax[d]);
        if (a->mass) {
            fprintf(F.fp, "\nMasses\n\n");
            for (int t = 1; t <= a->ntypes; t++) fprintf(F.fp, "%d %.17g\n", t, a->mass[t]);
        }
        // fields_data_atom = id molecule type q x ucgstate ucgl ucgml (+ image), UCG/atom_vec_ucg.cpp:87
        fprintf(F.fp, "\nAtoms # ucg\n\n");
        for (long long i = 0; i < a->n; i++)
            fprintf(F.fp, "%d %d %d %.17g %.17g %.17g %.17g %d %.17g %.17g %d %d %d\n", a->id[i],
                    a->molecule ? a->molecule[i] : 0, a->type[i], a->q ? a->q[i] : 0.0, a->x[3 * i], a->x[3 * i + 1],
                    a->x[3 * i + 2], a->ucgstate ? a->ucgstate[i] : 0, a->ucgl ? a->ucgl[i] : 0.0,
                    a->ucgml ? a->ucgml[i] : 0.0, a->image ? a->image[3 * i] : 0, a->image ? a->image[3 * i + 1] : 0,
                    a->image ? a->image[3 * i + 2] : 0);
        // fields_data_vel = id v ucgvl, UCG/atom_vec_ucg.cpp:90
        if (a->v) {
            fprintf(F.fp, "\nVelocities\n\n");
            for (long long i = 0; i < a->n; i++)
                fprintf(F.fp, "%d %.17g %.17g %.17g %.17g\n", a->id[i], a->v[3 * i], a->v[3 * i + 1], a->v[3 * i + 2],
                        a->ucgvl ? a->ucgvl[i] : 0.0);
        }
        F.finish(path);
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

}  // extern "C"
namespace {
const char *SECTIONS[] = {"Masses", "Atoms", "Velocities", "Pair Coeffs", "PairIJ Coeffs", "Bonds", "Angles", "Dihedrals",
                          "Impropers", "Bond Coeffs", "Angle Coeffs", "Dihedral Coeffs", "Improper Coeffs"};

// strips a trailing comment and surrounding blanks
std::string clean(const char *l) {
    std::string s(l);
    size_t h = s.find('#');
    if (h != std::string::npos) s.erase(h);
    while (!s.empty() && isspace((unsigned char) s.back())) s.pop_back();
    size_t b = 0;
    while (b < s.size() && isspace((unsigned char) s[b])) b++;
    return s.substr(b);
}

const char *section_of(const std::string &s) {
    for (const char *sec : SECTIONS)
        if (s == sec) return sec;
    return nullptr;
}

struct DataHeader {
    long long natoms = 0;
    int ntypes = 0;
    double lo[3] = {-0.5, -0.5, -0.5}, hi[3] = {0.5, 0.5, 0.5};
    bool has_vel = false;
};

// header keywords up to the first section name; returns that section line ("" at end of file)
std::string read_data_header(LineReader &R, DataHeader &H) {
    R.next();  // title line
    char *l;
    while ((l = R.next())) {
        std::string s = clean(l);
        if (s.empty()) continue;
        if (section_of(s)) return s;
        std::vector<std::string> w = split_ws(s);
        auto ends = [&](size_t k, const char *a0, const char *a1 = nullptr) {
            if (w.size() != k + (a1 ? 2 : 1)) return false;
            return w[k] == a0 && (!a1 || w[k + 1] == a1);
        };
        if (ends(1, "atoms")) H.natoms = atoll(w[0].c_str());
        else if (ends(1, "atom", "types")) H.ntypes = atoi(w[0].c_str());
        else if (ends(2, "xlo", "xhi")) { H.lo[0] = strtod(w[0].c_str(), nullptr); H.hi[0] = strtod(w[1].c_str(), nullptr); }
        else if (ends(2, "ylo", "yhi")) { H.lo[1] = strtod(w[0].c_str(), nullptr); H.hi[1] = strtod(w[1].c_str(), nullptr); }
        else if (ends(2, "zlo", "zhi")) { H.lo[2] = strtod(w[0].c_str(), nullptr); H.hi[2] = strtod(w[1].c_str(), nullptr); }
        else if (w.size() >= 2 && (w.back() == "types" || w.back() == "bonds" || w.back() == "angles" || w.back() == "dihedrals" ||
                                   w.back() == "impropers")) {
            if (atoll(w[0].c_str()) != 0 && w.back() != "types")
                fail(UCG_ERR_UNSUPPORTED, "Data file has %s: atom style ucg carries no topology (UCG/atom_vec_ucg.cpp:110-112)", w.back().c_str());
        } else if (w.size() == 4 && w[3] == "yz") fail(UCG_ERR_UNSUPPORTED, "Triclinic data files are not supported");
        else fail(UCG_ERR_INPUT, "Unknown identifier in data file: %s", s.c_str());
    }
    return "";
}
}  // namespace
extern "C" {

int ucg_io_data_header(const char *path, long long *natoms, int *ntypes, double *boxlo, double *boxhi, int *has_velocities,
                       char *err, int errcap) {
    try {
        File F(path, "r");
        LineReader R(F.fp);
        DataHeader H;
        std::string sec = read_data_header(R, H);
        char *l;
        while (!sec.empty()) {
            if (sec == "Velocities") H.has_vel = true;
            sec.clear();
            while ((l = R.next())) {
                std::string s = clean(l);
                if (section_of(s)) {
                    sec = s;
                    break;
                }
            }
        }
        if (natoms) *natoms = H.natoms;
        if (ntypes) *ntypes = H.ntypes;
        for (int d = 0; d < 3; d++) {
            if (boxlo) boxlo[d] = H.lo[d];
            if (boxhi) boxhi[d] = H.hi[d];
        }
        if (has_velocities) *has_velocities = H.has_vel ? 1 : 0;
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

/* a->n / a->ntypes give the capacities of the caller's arrays (from ucg_io_data_header); atoms are stored in file
 * order; Velocities are matched by ID.  data_atom_post is applied (ucgl, ucgstate clamped, ucgp = -1). */
int ucg_io_read_data(const char *path, ucg_io_atoms *a, char *err, int errcap) {
    try {
        if (!path || !a || !a->id || !a->type || !a->x) fail(UCG_ERR_INVALID, "ucg_io_read_data: id, type and x are required");
        File F(path, "r");
        LineReader R(F.fp);
        DataHeader H;
        std::string sec = read_data_header(R, H);
        if (H.natoms > a->n || H.ntypes > a->ntypes) fail(UCG_ERR_INVALID, "ucg_io_read_data: arrays too small (%lld atoms, %d types)", H.natoms, H.ntypes);
        a->n = H.natoms;
        a->ntypes = H.ntypes;
        for (int d = 0; d < 3; d++) {
            a->boxlo[d] = H.lo[d];
            a->boxhi[d] = H.hi[d];
        }
        bool atoms_done = false;
        std::unordered_map<int, long long> map;
        char *l;
        auto next_data_line = [&]() -> char * {
            while ((l = R.next())) {
                const char *p = l;
                while (*p && isspace((unsigned char) *p)) p++;
                if (*p) return l;
            }
            return nullptr;
        };
        while (!sec.empty()) {
            std::string cur = sec;
            sec.clear();
            if (cur == "Masses") {
                for (int t = 0; t < H.ntypes; t++) {
                    if (!next_data_line()) fail(UCG_ERR_INPUT, "Unexpected end of data file");
                    int ty;
                    double m;
                    if (sscanf(l, "%d %lf", &ty, &m) != 2 || ty < 1 || ty > H.ntypes) fail(UCG_ERR_INPUT, "Invalid Masses line in data file: %s", l);
                    if (m <= 0.0) fail(UCG_ERR_INPUT, "Invalid mass value");
                    if (a->mass) a->mass[ty] = m;
                }
            } else if (cur == "Atoms") {
                for (long long i = 0; i < H.natoms; i++) {
                    if (!next_data_line()) fail(UCG_ERR_INPUT, "Unexpected end of data file");
                    std::vector<std::string> w = split_ws(clean(l));
                    if (w.size() != 10 && w.size() != 13) fail(UCG_ERR_INPUT, "Incorrect atom format in data file: %s", l);
                    a->id[i] = atoi(w[0].c_str());
                    if (a->molecule) a->molecule[i] = atoi(w[1].c_str());
                    a->type[i] = atoi(w[2].c_str());
                    if (a->type[i] < 1 || a->type[i] > H.ntypes) fail(UCG_ERR_INPUT, "Invalid atom type in Atoms section of data file");
                    if (a->q) a->q[i] = strtod(w[3].c_str(), nullptr);
                    for (int d = 0; d < 3; d++) a->x[3 * i + d] = strtod(w[4 + d].c_str(), nullptr);
                    int st = atoi(w[7].c_str());
                    double lam = strtod(w[8].c_str(), nullptr);
                    // data_atom_post, UCG/atom_vec_ucg.cpp:155-169
                    if (lam < 0) lam = 0.;
                    else if (lam > 1) lam = 1.;
                    if (st < 0) st = 0;
                    else if (st > 1) st = 1;
                    if (a->ucgstate) a->ucgstate[i] = st;
                    if (a->ucgl) a->ucgl[i] = lam;
                    if (a->ucgml) a->ucgml[i] = strtod(w[9].c_str(), nullptr);
                    if (a->ucgp) a->ucgp[i] = -1.0;
                    if (a->image)
                        for (int d = 0; d < 3; d++) a->image[3 * i + d] = w.size() == 13 ? atoi(w[10 + d].c_str()) : 0;
                    if (a->v) a->v[3 * i] = a->v[3 * i + 1] = a->v[3 * i + 2] = 0.0;
                    if (a->ucgvl) a->ucgvl[i] = 0.0;
                    map[a->id[i]] = i;
                }
                if ((long long) map.size() != H.natoms) fail(UCG_ERR_INPUT, "Duplicate atom IDs in data file");
                atoms_done = true;
            } else if (cur == "Velocities") {
                if (!atoms_done) fail(UCG_ERR_INPUT, "Must read Atoms before Velocities");
                for (long long i = 0; i < H.natoms; i++) {
                    if (!next_data_line()) fail(UCG_ERR_INPUT, "Unexpected end of data file");
                    int id;
                    double v0, v1, v2, vl;
                    if (sscanf(l, "%d %lf %lf %lf %lf", &id, &v0, &v1, &v2, &vl) != 5) fail(UCG_ERR_INPUT, "Incorrect velocity format in data file: %s", l);
                    auto it = map.find(id);
                    if (it == map.end()) fail(UCG_ERR_INPUT, "Invalid atom ID in Velocities section of data file");
                    long long m = it->second;
                    if (a->v) { a->v[3 * m] = v0; a->v[3 * m + 1] = v1; a->v[3 * m + 2] = v2; }
                    if (a->ucgvl) a->ucgvl[m] = vl;
                }
            } else {
                // coefficient sections are input-script business here: skip to the next section
            }
            while ((l = R.next())) {
                std::string s = clean(l);
                if (section_of(s)) {
                    sec = s;
                    break;
                }
                if (!s.empty() && (cur == "Masses" || cur == "Atoms" || cur == "Velocities"))
                    fail(UCG_ERR_INPUT, "Unknown identifier in data file: %s", s.c_str());
            }
        }
        if (!atoms_done && H.natoms) fail(UCG_ERR_INPUT, "No Atoms section in data file");
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

// ---------------------------------------------------------------------------------- restart container

}  // extern "C"
namespace {
const char RST_MAGIC[8] = {'U', 'C', 'G', 'R', 'S', 'T', '0', '1'};
struct RstHeader {
    char magic[8];
    int64_t timestep, natoms;
    int32_t ntypes, present;  // bit mask of optional arrays
    double lo[3], hi[3];
};
enum { P_MOL = 1, P_V = 2, P_Q = 4, P_IMAGE = 8, P_STATE = 16, P_L = 32, P_ML = 64, P_VL = 128, P_P = 256, P_MASS = 512 };

void read_rst_header(FILE *fp, RstHeader &H) {
    if (fread(&H, 1, sizeof H, fp) != sizeof H || memcmp(H.magic, RST_MAGIC, 8) != 0)
        fail(UCG_ERR_INPUT, "Restart file is not a libucg_hip restart (UCGRST01)");
}
}  // namespace
extern "C" {

int ucg_io_write_restart(const char *path, long long timestep, const ucg_io_atoms *a, char *err, int errcap) {
    try {
        if (!path || !a || !a->id || !a->type || !a->x) fail(UCG_ERR_INVALID, "ucg_io_write_restart: id, type and x are required");
        File F(path, "wb");
        RstHeader H;
        memcpy(H.magic, RST_MAGIC, 8);
        H.timestep = timestep;
        H.natoms = a->n;
        H.ntypes = a->ntypes;
        H.present = (a->molecule ? P_MOL : 0) | (a->v ? P_V : 0) | (a->q ? P_Q : 0) | (a->image ? P_IMAGE : 0) |
                    (a->ucgstate ? P_STATE : 0) | (a->ucgl ? P_L : 0) | (a->ucgml ? P_ML : 0) | (a->ucgvl ? P_VL : 0) |
                    (a->ucgp ? P_P : 0) | (a->mass ? P_MASS : 0);
        for (int d = 0; d < 3; d++) {
            H.lo[d] = a->boxlo[d];
            H.hi[d] = a->boxhi[d];
        }
        auto put = [&](const void *p, size_t bytes) {
            if (p && fwrite(p, 1, bytes, F.fp) != bytes) fail(UCG_ERR_INPUT, "Restart file write error");
        };
        size_t n = (size_t) a->n;
        put(&H, sizeof H);
        put(a->id, 4 * n); put(a->type, 4 * n); put(a->x, 24 * n);
        put(a->molecule, 4 * n); put(a->v, 24 * n); put(a->q, 8 * n); put(a->image, 12 * n);
        // fields_restart of the atom style, UCG/atom_vec_ucg.cpp:85
        put(a->ucgstate, 4 * n); put(a->ucgl, 8 * n); put(a->ucgml, 8 * n); put(a->ucgvl, 8 * n); put(a->ucgp, 8 * n);
        put(a->mass, 8 * (size_t) (a->ntypes + 1));
        F.finish(path);
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

int ucg_io_restart_header(const char *path, long long *timestep, long long *natoms, int *ntypes, double *boxlo,
                          double *boxhi, char *err, int errcap) {
    try {
        File F(path, "rb");
        RstHeader H;
        read_rst_header(F.fp, H);
        if (timestep) *timestep = H.timestep;
        if (natoms) *natoms = H.natoms;
        if (ntypes) *ntypes = H.ntypes;
        for (int d = 0; d < 3; d++) {
            if (boxlo) boxlo[d] = H.lo[d];
            if (boxhi) boxhi[d] = H.hi[d];
        }
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

int ucg_io_read_restart(const char *path, ucg_io_atoms *a, long long *timestep, char *err, int errcap) {
    try {
        if (!path || !a) fail(UCG_ERR_INVALID, "ucg_io_read_restart: null argument");
        File F(path, "rb");
        RstHeader H;
        read_rst_header(F.fp, H);
        if (H.natoms > a->n || H.ntypes > a->ntypes) fail(UCG_ERR_INVALID, "ucg_io_read_restart: arrays too small (%lld atoms, %d types)", (long long) H.natoms, H.ntypes);
        a->n = H.natoms;
        a->ntypes = H.ntypes;
        for (int d = 0; d < 3; d++) {
            a->boxlo[d] = H.lo[d];
            a->boxhi[d] = H.hi[d];
        }
        size_t n = (size_t) H.natoms;
        std::vector<char> sink;
        auto get = [&](void *p, size_t bytes, bool present) {
            if (!present) return;
            void *dst = p;
            if (!dst) {
                sink.resize(bytes);
                dst = sink.data();
            }
            if (fread(dst, 1, bytes, F.fp) != bytes) fail(UCG_ERR_INPUT, "Restart file is truncated");
        };
        get(a->id, 4 * n, true); get(a->type, 4 * n, true); get(a->x, 24 * n, true);
        get(a->molecule, 4 * n, H.present & P_MOL); get(a->v, 24 * n, H.present & P_V); get(a->q, 8 * n, H.present & P_Q);
        get(a->image, 12 * n, H.present & P_IMAGE); get(a->ucgstate, 4 * n, H.present & P_STATE);
        get(a->ucgl, 8 * n, H.present & P_L); get(a->ucgml, 8 * n, H.present & P_ML); get(a->ucgvl, 8 * n, H.present & P_VL);
        get(a->ucgp, 8 * n, H.present & P_P); get(a->mass, 8 * (size_t) (H.ntypes + 1), H.present & P_MASS);
        if (timestep) *timestep = H.timestep;
        return UCG_OK;
    } catch (const IoError &e) {
        return report(e, err, errcap);
    }
}

}  // extern "C"
