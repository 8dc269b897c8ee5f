// lammps_api_decl.h -- DECLARATIONS ONLY of the part of the upstream LAMMPS C++ API that the USER-UCG/GPU glue
// (lammps-ucg-dev_amd/lammps/*.cpp) uses, written from the public LAMMPS developer documentation so that a C++
// compiler can check OUR glue (types, overrides, argument lists) in a container that has no LAMMPS tree:
//     g++ -std=c++17 -fsyntax-only -I tests/lammps_api_decl -I include lammps-ucg-dev_amd/lammps/<file>.cpp
// (tests/test_glue_compiles.py).  Nothing here is an implementation, nothing is linked, and it is NOT used to build
// anything of /root/reference (which stays unbuildable here: DESIGN.md section 2).  The real build compiles the glue
// inside a LAMMPS source tree against the real headers.
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_INT 1
#define MPI_BYTE 2
#define MPI_LONG_LONG 3
#define MPI_DOUBLE 4
#define MPI_SUM 1
#define MPI_MAX 2
#define MPI_MIN 3
#define MPI_SUCCESS 0
#define MPI_IN_PLACE ((void *) 1)
#define MPI_COMM_NULL 0
int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm);
int MPI_Comm_split(MPI_Comm, int color, int key, MPI_Comm *newcomm);
int MPI_Comm_free(MPI_Comm *);
int MPI_Alltoall(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, MPI_Comm);
int MPI_Alltoallv(const void *, const int *, const int *, MPI_Datatype, void *, const int *, const int *, MPI_Datatype, MPI_Comm);
int MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm);

namespace LAMMPS_NS {

typedef int tagint;
typedef int imageint;
typedef int64_t bigint;
#define IMGMASK 1023
#define IMGMAX 512
#define IMGBITS 10
#define IMG2BITS 20
#define NEIGHMASK 0x1FFFFFFF
#define SBBITS 30
#define FLERR __FILE__, __LINE__

class LAMMPS;
class Memory;
class Error;
class Atom;
class AtomVec;
class Update;
class Neighbor;
class NeighList;
class Comm;
class Domain;
class Force;
class Modify;
class Output;
class Pair;
class Fix;
class Compute;
class Group;
class Timer;

class Pointers {
 public:
  Pointers(LAMMPS *);
  virtual ~Pointers() = default;

 protected:
  LAMMPS *lmp;
  Memory *&memory;
  Error *&error;
  Atom *&atom;
  Update *&update;
  Neighbor *&neighbor;
  Comm *&comm;
  Domain *&domain;
  Force *&force;
  Modify *&modify;
  Output *&output;
  Group *&group;
  Timer *&timer;
  MPI_Comm &world;
  FILE *&screen;
  FILE *&logfile;
};

class LAMMPS {
 public:
  Memory *memory;
  Error *error;
  Atom *atom;
  Update *update;
  Neighbor *neighbor;
  Comm *comm;
  Domain *domain;
  Force *force;
  Modify *modify;
  Output *output;
  Group *group;
  Timer *timer;
  MPI_Comm world;
  FILE *screen, *logfile;
};

class Error : protected Pointers {
 public:
  Error(LAMMPS *);
  [[noreturn]] void all(const std::string &file, int line, const std::string &msg);
  [[noreturn]] void one(const std::string &file, int line, const std::string &msg);
  template <typename... Args> [[noreturn]] void all(const std::string &file, int line, const std::string &fmt, Args &&...args);
  template <typename... Args> [[noreturn]] void all(const std::string &file, int line, int errptr, const std::string &fmt, Args &&...args);
  template <typename... Args> [[noreturn]] void one(const std::string &file, int line, const std::string &fmt, Args &&...args);
  void warning(const std::string &file, int line, const std::string &msg);
  template <typename... Args> void warning(const std::string &file, int line, const std::string &fmt, Args &&...args);
};

class Memory : protected Pointers {
 public:
  Memory(LAMMPS *);
  template <typename T> T **create(T **&array, int n1, int n2, const char *name);
  template <typename T> void destroy(T **&array);
};

namespace utils {
bool strmatch(const std::string &text, const std::string &pattern);
double numeric(const char *file, int line, const std::string &str, bool do_abort, LAMMPS *lmp);
int inumeric(const char *file, int line, const std::string &str, bool do_abort, LAMMPS *lmp);
void sfread(const char *file, int line, void *s, size_t size, size_t num, FILE *fp, const char *filename, Error *error);
void missing_cmd_args(const std::string &file, int line, const std::string &cmd, Error *error);
char *strdup(const std::string &text);
void logmesg(LAMMPS *lmp, const std::string &mesg);
}    // namespace utils

class Atom : protected Pointers {
 public:
  enum { DOUBLE, INT, BIGINT };
  enum { ATOMIC = 0, MOLECULAR = 1, TEMPLATE = 2 };
  enum { MAP_NONE = 0, MAP_ARRAY = 1, MAP_HASH = 2, MAP_YES = 3 };
  Atom(LAMMPS *);
  int nlocal, nghost, ntypes, nmax;
  bigint natoms;
  int map_style;
  int molecule_flag, q_flag;
  AtomVec *avec;
  tagint *tag;
  int *type, *mask;
  tagint *molecule;
  double **x, **v, **f;
  double *q;
  imageint *image;
  double *mass, *rmass;
  int *num_bond, *num_angle, *num_dihedral, *num_improper;
  int **nspecial;
  void add_peratom(const std::string &name, void *address, int datatype, int cols, int threadflag = 0);
  int map(tagint global);
  void map_init(int check = 1);
  void map_set();
  void setup();
};

class AtomVec : protected Pointers {
 public:
  enum { PER_ATOM = 0, PER_TYPE = 1 };
  AtomVec(LAMMPS *);
  int molecular;
  int bonds_allow, angles_allow, dihedrals_allow, impropers_allow;
  int mass_type;
  int forceclearflag;
  std::vector<std::string> fields_grow, fields_copy, fields_comm, fields_comm_vel, fields_reverse, fields_border,
      fields_border_vel, fields_exchange, fields_restart, fields_create, fields_data_atom, fields_data_vel;
  virtual void grow(int);
  virtual void grow_pointers() {}
  virtual void force_clear(int, size_t) {}
  virtual void data_atom_post(int) {}
  virtual int property_atom(const std::string &) { return -1; }
  virtual void pack_property_atom(int, double *, int, int) {}

 protected:
  void setup_fields();
};

class Integrate : protected Pointers {
 public:
  Integrate(LAMMPS *, int, char **);
  virtual void init();
  virtual void setup(int flag) = 0;
  virtual void setup_minimal(int) = 0;
  virtual void run(int) = 0;
  virtual void force_clear() = 0;
  virtual void cleanup() {}
  virtual void reset_dt() {}
  virtual double memory_usage() { return 0; }

 protected:
  int eflag, vflag;
  void ev_setup();
  void ev_set(bigint);
};
class Respa : public Integrate {
 public:
  Respa(LAMMPS *, int, char **);
  int nlevels;
  double *step;
  void copy_f_flevel(int);
  void copy_flevel_f(int);
};

class Update : protected Pointers {
 public:
  Update(LAMMPS *);
  double dt;
  bigint ntimestep, beginstep, endstep, firststep, laststep;
  int setupflag, whichflag;
  char *integrate_style;
  Integrate *integrate;
  void update_time();
};

class Force : protected Pointers {
 public:
  Force(LAMMPS *);
  double boltz, ftm2v, mvv2e;
  double special_lj[4];
  Pair *pair;
};

class Modify : protected Pointers {
 public:
  Modify(LAMMPS *);
  int nfix;
  Fix **fix;
  Compute *get_compute_by_id(const std::string &) const;
  void post_run();
  void setup(int);
};

class Compute : protected Pointers {
 public:
  Compute(LAMMPS *, int, char **);
  char *id;
  int igroup;
  int tempflag, tempbias;
  virtual double compute_scalar() { return 0.0; }
};

class Group : protected Pointers {
 public:
  Group(LAMMPS *);
  char **names;
};

class Timer : protected Pointers {
 public:
  enum ttype { RESET = -2, START = -1, TOTAL = 0, PAIR, BOND, KSPACE, NEIGH, COMM, MODIFY, OUTPUT, SYNC, ALL };
  Timer(LAMMPS *);
  void stamp();
  void stamp(enum ttype);
  void init_timeout();
};

class Domain : protected Pointers {
 public:
  Domain(LAMMPS *);
  double boxlo[3], boxhi[3], prd[3];
  int triclinic;
  int xperiodic, yperiodic, zperiodic;
  void pbc();
  void reset_box();
  void box_too_small_check();
};

class Output : protected Pointers {
 public:
  Output(LAMMPS *);
  bigint next;    // next timestep with any output (thermo, dump, restart)
  bigint next_thermo;
  void setup(int memflag = 1);
  void write(bigint);
};

class Comm : protected Pointers {
 public:
  Comm(LAMMPS *);
  enum { LAYOUT_UNIFORM, LAYOUT_NONUNIFORM, LAYOUT_TILED };
  int me, nprocs;
  int layout;
  int procgrid[3], myloc[3];
  virtual void forward_comm(Pair *, int size = 0);
};

class NeighList : protected Pointers {
 public:
  NeighList(LAMMPS *);
  int inum;
  int *ilist, *numneigh;
  int **firstneigh;
};

namespace NeighConst {
enum { REQ_DEFAULT = 0, REQ_FULL = 1 << 0 };
}
class NeighRequest;
class Neighbor : protected Pointers {
 public:
  Neighbor(LAMMPS *);
  bigint ncalls, lastcall;
  int every, delay, dist_check, ago;
  double skin;
  NeighRequest *add_request(Pair *, int flags = 0);
};

class Pair : protected Pointers {
 public:
  Pair(LAMMPS *);
  double eng_vdwl;
  double virial[6];
  int no_virial_fdotr_compute;
  int comm_forward;
  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
  virtual void coeff(int, char **) = 0;
  virtual void init_style();
  virtual double init_one(int, int) { return 0.0; }
  virtual double single(int, int, int, int, double, double, double, double &) { return 0.0; }
  virtual void *extract(const char *, int &) { return nullptr; }
  virtual void write_restart(FILE *) {}
  virtual void read_restart(FILE *) {}
  virtual void write_restart_settings(FILE *) {}
  virtual void read_restart_settings(FILE *) {}
  virtual int pack_forward_comm(int, int *, double *, int, int *) { return 0; }
  virtual void unpack_forward_comm(int, int, double *) {}

 protected:
  int allocated, copymode;
  int **setflag;
  double **cutsq;
  NeighList *list;
  int eflag_global, vflag_global;
  int ewaldflag, pppmflag, msmflag, dispersionflag, tip4pflag;
  void ev_init(int eflag, int vflag, int alloc = 1);
};

namespace FixConst {
enum {
  INITIAL_INTEGRATE = 1 << 0, POST_INTEGRATE = 1 << 1, PRE_EXCHANGE = 1 << 2, PRE_NEIGHBOR = 1 << 3,
  POST_NEIGHBOR = 1 << 4, PRE_FORCE = 1 << 5, PRE_REVERSE = 1 << 6, POST_FORCE = 1 << 7, FINAL_INTEGRATE = 1 << 8,
  END_OF_STEP = 1 << 9, POST_RUN = 1 << 10, INITIAL_INTEGRATE_RESPA = 1 << 11, POST_INTEGRATE_RESPA = 1 << 12,
  PRE_FORCE_RESPA = 1 << 13, POST_FORCE_RESPA = 1 << 14, FINAL_INTEGRATE_RESPA = 1 << 15, MIN_PRE_EXCHANGE = 1 << 16, MIN_PRE_NEIGHBOR = 1 << 17,
  MIN_POST_NEIGHBOR = 1 << 18, MIN_PRE_FORCE = 1 << 19, MIN_PRE_REVERSE = 1 << 20, MIN_POST_FORCE = 1 << 21
};
}
class Fix : protected Pointers {
 public:
  Fix(LAMMPS *, int, char **);
  ~Fix() override = default;
  char *id, *style;
  int igroup, groupbit;
  int nevery, time_integrate, dynamic_group_allow, force_reneighbor;
  bigint next_reneighbor;
  int scalar_flag, vector_flag, size_vector, global_freq, extscalar, extvector;
  virtual int setmask() = 0;
  virtual void init() {}
  virtual void setup(int) {}
  virtual void initial_integrate(int) {}
  virtual void post_force(int) {}
  virtual void final_integrate() {}
  virtual void initial_integrate_respa(int, int, int) {}
  virtual void final_integrate_respa(int, int) {}
  virtual void end_of_step() {}
  virtual void pre_exchange() {}
  virtual void post_run() {}
  virtual void post_force_respa(int, int, int) {}
  virtual void reset_target(double) {}
  virtual int modify_param(int, char **) { return 0; }
  virtual void reset_dt() {}
  virtual double compute_scalar() { return 0.0; }
  virtual double compute_vector(int) { return 0.0; }
  virtual void *extract(const char *, int &) { return nullptr; }
};

}    // namespace LAMMPS_NS
