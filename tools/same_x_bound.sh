cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "# same-x fusion bound: the 3x3x3 conv launches of the supernet's C = 4 / 8 levels with their halo fill (as built) and WITHOUT it (-DVOX_NO_LOAD: the"
echo "# LDS tile is read as it lies) -- the difference is everything a second conv reading the same input tile could save"
python3 tools/conv_ab.py 4 64 1 2 4 64 2 2 8 32 1 2 8 32 2 2 8 64 1 2 8 64 2 2 2>&1 | grep -v amdgpu.ids
N3D_LIB=$GRAFT_REPO_ROOT/tools/build/libn3d_VOX_NO_LOAD.so python3 tools/conv_ab.py 4 64 1 2 4 64 2 2 8 32 1 2 8 32 2 2 8 64 1 2 8 64 2 2 2>&1 | grep -v amdgpu.ids
