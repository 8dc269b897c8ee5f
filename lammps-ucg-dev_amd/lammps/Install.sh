# Install/unInstall USER-UCG/GPU into a LAMMPS src/ directory (same mechanism as the
# reference's UCG/Install.sh:1-30): mode = 0/1/2 for uninstall/install/update.
mode=$1
action () {
  if (test $mode = 0) then
    rm -f ../$1
  elif (! cmp -s $1 ../$1) then
    cp $1 ../$1
  fi
}
for file in *.cpp *.h; do   # atom_vec_ucg_gpu, pair_table_ucg_gpu, fix_ucg_gpu, verlet_ucg_gpu
  test -f ${file} && action $file
done
