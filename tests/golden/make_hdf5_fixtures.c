/* Writes the HDF5 reader fixtures of tests/golden/hdf5/ with the REAL libhdf5 (1.10.6, /opt/conda in the build image), in the shapes
 * h5py / Keras produce (keras/engine/saving.py: save_weights_to_hdf5_group, _save_model; reference call sites T1:1046-1047, T1:1079).
 * Run by tests/golden/make_hdf5_fixtures.sh; the files are committed, this program only documents how they were made.
 *
 * Every float dataset holds val(seed, i) = ((i + 977*seed) * 2654435761 mod 2^32 >> 8) / 2^24 - 0.5 over its flat index i: exact in
 * float32, so tests/test_hdf5_pinned.py recomputes the expected contents instead of storing them.
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "HDF5 call failed: %s (line %d)\n", #x, __LINE__); exit(1); } } while (0)

static float val(unsigned seed, unsigned i) {
    unsigned h = (i + 977u * seed) * 2654435761u;
    return (float)(h >> 8) / 16777216.0f - 0.5f;
}

static void attr_fixed_str(hid_t loc, const char* name, const char* s) {          /* h5py: attrs[name] = b"..." (numpy bytes_) */
    hid_t t = H5Tcopy(H5T_C_S1), sp = H5Screate(H5S_SCALAR);
    CHECK(H5Tset_size(t, strlen(s))); CHECK(H5Tset_strpad(t, H5T_STR_NULLPAD));
    hid_t a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT); CHECK(a);
    CHECK(H5Awrite(a, t, s)); H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}

static void attr_empty_f64(hid_t loc, const char* name) {   /* h5py: attrs[name] = [] -> a float64 array of shape (0,) (weight-less layers) */
    hsize_t d = 0; hid_t sp = H5Screate_simple(1, &d, NULL); double dummy = 0;
    hid_t a = H5Acreate2(loc, name, H5T_IEEE_F64LE, sp, H5P_DEFAULT, H5P_DEFAULT); CHECK(a);
    CHECK(H5Awrite(a, H5T_NATIVE_DOUBLE, &dummy)); H5Aclose(a); H5Sclose(sp);
}

static void attr_fixed_str_array(hid_t loc, const char* name, const char** s, int n) {   /* h5py: attrs[name] = np.array([...], 'S') */
    if (n == 0) { attr_empty_f64(loc, name); return; }
    size_t w = 1;
    for (int i = 0; i < n; i++) if (strlen(s[i]) > w) w = strlen(s[i]);
    char* buf = calloc((size_t)n, w);
    for (int i = 0; i < n; i++) memcpy(buf + (size_t)i * w, s[i], strlen(s[i]));
    hid_t t = H5Tcopy(H5T_C_S1); hsize_t d = (hsize_t)n; hid_t sp = H5Screate_simple(1, &d, NULL);
    CHECK(H5Tset_size(t, w)); CHECK(H5Tset_strpad(t, H5T_STR_NULLPAD));
    hid_t a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT); CHECK(a);
    CHECK(H5Awrite(a, t, buf)); H5Aclose(a); H5Sclose(sp); H5Tclose(t); free(buf);
}

static void attr_vlen_str(hid_t loc, const char* name, const char* s) {           /* h5py 3: attrs[name] = "..." (str) */
    hid_t t = H5Tcopy(H5T_C_S1), sp = H5Screate(H5S_SCALAR);
    CHECK(H5Tset_size(t, H5T_VARIABLE)); CHECK(H5Tset_cset(t, H5T_CSET_UTF8));
    hid_t a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT); CHECK(a);
    CHECK(H5Awrite(a, t, &s)); H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}

static void attr_vlen_str_array(hid_t loc, const char* name, const char** s, int n) {    /* h5py 3: a list of str */
    if (n == 0) { attr_empty_f64(loc, name); return; }
    hid_t t = H5Tcopy(H5T_C_S1); hsize_t d = (hsize_t)n; hid_t sp = H5Screate_simple(1, &d, NULL);
    CHECK(H5Tset_size(t, H5T_VARIABLE)); CHECK(H5Tset_cset(t, H5T_CSET_UTF8));
    hid_t a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT); CHECK(a);
    CHECK(H5Awrite(a, t, s)); H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}

/* dataset of val(seed, .) with the given file type; h5py creates intermediate groups for "conv2d_1/kernel:0" */
static void dset_f32(hid_t loc, const char* name, int rank, const hsize_t* dims, unsigned seed, hid_t filetype, hid_t dcpl) {
    size_t n = 1;
    for (int i = 0; i < rank; i++) n *= dims[i];
    float* buf = malloc(sizeof(float) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) buf[i] = val(seed, (unsigned)i);
    hid_t sp = rank ? H5Screate_simple(rank, dims, NULL) : H5Screate(H5S_SCALAR);
    hid_t lcpl = H5Pcreate(H5P_LINK_CREATE); CHECK(H5Pset_create_intermediate_group(lcpl, 1));
    hid_t d = H5Dcreate2(loc, name, filetype, sp, lcpl, dcpl, H5P_DEFAULT); CHECK(d);
    CHECK(H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf));
    H5Dclose(d); H5Pclose(lcpl); H5Sclose(sp); free(buf);
}

static hid_t dcpl_h5py(void) {                    /* h5py creates datasets with track_times off */
    hid_t p = H5Pcreate(H5P_DATASET_CREATE); CHECK(H5Pset_obj_track_times(p, 0)); return p;
}

/* the two layers every fixture carries: a conv (kernel 3x3x1x4 + bias 4) and a BatchNorm (4 vectors of 4), plus a weight-less layer */
static void keras_layers(hid_t g, int vlen_names, hid_t filetype, hid_t dcpl) {
    const char* layers[] = {"input_1", "conv2d_1", "batch_normalization_1", "a_layer_with_a_rather_long_name_1"};
    const char* wconv[] = {"conv2d_1/kernel:0", "conv2d_1/bias:0"};
    const char* wbn[] = {"batch_normalization_1/gamma:0", "batch_normalization_1/beta:0", "batch_normalization_1/moving_mean:0",
                         "batch_normalization_1/moving_variance:0"};
    const char* wlong[] = {"a_layer_with_a_rather_long_name_1/kernel:0"};
    if (vlen_names) attr_vlen_str_array(g, "layer_names", layers, 4); else attr_fixed_str_array(g, "layer_names", layers, 4);
    if (vlen_names) { attr_vlen_str(g, "backend", "tensorflow"); attr_vlen_str(g, "keras_version", "2.4.0"); }
    else { attr_fixed_str(g, "backend", "tensorflow"); attr_fixed_str(g, "keras_version", "2.3.1"); }
    for (int l = 0; l < 4; l++) {
        hid_t lg = H5Gcreate2(g, layers[l], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); CHECK(lg);
        const char** wn = l == 1 ? wconv : l == 2 ? wbn : l == 3 ? wlong : NULL;
        int nw = l == 1 ? 2 : l == 2 ? 4 : l == 3 ? 1 : 0;
        if (vlen_names) attr_vlen_str_array(lg, "weight_names", wn, nw); else attr_fixed_str_array(lg, "weight_names", wn, nw);
        for (int k = 0; k < nw; k++) {
            hsize_t d4[4] = {3, 3, 1, 4}, d1[1] = {4}, d2[2] = {5, 7};
            if (l == 1 && k == 0) dset_f32(lg, wn[k], 4, d4, 10u * l + k, filetype, dcpl);
            else if (l == 3) dset_f32(lg, wn[k], 2, d2, 10u * l + k, filetype, dcpl);
            else dset_f32(lg, wn[k], 1, d1, 10u * l + k, filetype, dcpl);
        }
        H5Gclose(lg);
    }
}

static const char* MODEL_CONFIG =
    "{\"class_name\": \"Model\", \"config\": {\"name\": \"model_1\", \"layers\": [{\"name\": \"input_1\", \"class_name\": \"InputLayer\"}, "
    "{\"name\": \"conv2d_1\", \"class_name\": \"Conv2D\", \"config\": {\"filters\": 4, \"kernel_size\": [3, 3], \"padding\": \"same\"}}]}}";

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s OUTDIR\n", argv[0]); return 2; }
    char path[4096];
    hid_t dcpl = dcpl_h5py();

    /* 1. model.save_weights as Keras 2.3 + h5py 2.x wrote it: libver earliest, fixed-length NULL-padded byte strings */
    snprintf(path, sizeof path, "%s/weights_h5py2_earliest.h5", argv[1]);
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT); CHECK(f);
    keras_layers(f, 0, H5T_IEEE_F32LE, dcpl);
    CHECK(H5Fclose(f));

    /* 2. model.save (full model) as tf.keras + h5py 3 writes it: variable-length UTF-8 strings (global heap), model_weights/ and
     *    optimizer_weights/ groups, an int64 scalar and a float64 dataset; attributes added after the children exist (continuation blocks) */
    snprintf(path, sizeof path, "%s/fullmodel_h5py3_vlen.h5", argv[1]);
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT); CHECK(f);
    hid_t mw = H5Gcreate2(f, "model_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); CHECK(mw);
    keras_layers(mw, 1, H5T_IEEE_F32LE, dcpl);
    H5Gclose(mw);
    hid_t ow = H5Gcreate2(f, "optimizer_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); CHECK(ow);
    { const char* on[] = {"Adam/iterations:0", "Adam/conv2d_1/kernel/m:0"};
      attr_vlen_str_array(ow, "weight_names", on, 2);
      long long it = 1234567890123LL; hid_t sp = H5Screate(H5S_SCALAR);
      hid_t lcpl = H5Pcreate(H5P_LINK_CREATE); CHECK(H5Pset_create_intermediate_group(lcpl, 1));
      hid_t d = H5Dcreate2(ow, on[0], H5T_STD_I64LE, sp, lcpl, dcpl, H5P_DEFAULT); CHECK(d);
      CHECK(H5Dwrite(d, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, &it)); H5Dclose(d); H5Sclose(sp); H5Pclose(lcpl);
      hsize_t d4[4] = {3, 3, 1, 4}; dset_f32(ow, on[1], 4, d4, 77, H5T_IEEE_F64LE, dcpl); }
    H5Gclose(ow);
    attr_vlen_str(f, "keras_version", "2.4.0"); attr_vlen_str(f, "backend", "tensorflow");
    attr_vlen_str(f, "model_config", MODEL_CONFIG);
    attr_vlen_str(f, "training_config", "{\"loss\": \"bce_dice_loss\", \"optimizer_config\": {\"class_name\": \"Adam\", \"config\": {\"lr\": 0.0005}}}");
    CHECK(H5Fclose(f));

    /* 3. the same logical weights file from a writer that set libver='latest': superblock v3, version-2 object headers, compact link
     *    messages (<= 8 links per group), version-3 attribute messages; one compact-layout dataset and big-endian floats */
    snprintf(path, sizeof path, "%s/weights_latest_compact_be.h5", argv[1]);
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS); CHECK(H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST));
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl); CHECK(f);
    hid_t dc = dcpl_h5py(); CHECK(H5Pset_layout(dc, H5D_COMPACT));
    keras_layers(f, 0, H5T_IEEE_F32BE, dc);
    H5Pclose(dc); CHECK(H5Fclose(f));

    /* 4. default C-library dataset creation (modification-time messages kept) with the object-time tracking left on */
    snprintf(path, sizeof path, "%s/weights_tracked_times.h5", argv[1]);
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT); CHECK(f);
    keras_layers(f, 0, H5T_IEEE_F32LE, H5P_DEFAULT);
    CHECK(H5Fclose(f));

    /* 5. what the reader must refuse by name: a chunked + deflate dataset */
    snprintf(path, sizeof path, "%s/refused_chunked_deflate.h5", argv[1]);
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT); CHECK(f);
    { hid_t p = dcpl_h5py(); hsize_t ch[2] = {4, 4}, d2[2] = {8, 8}; CHECK(H5Pset_chunk(p, 2, ch)); CHECK(H5Pset_deflate(p, 4));
      dset_f32(f, "x", 2, d2, 5, H5T_IEEE_F32LE, p); H5Pclose(p); }
    CHECK(H5Fclose(f));

    /* 6. ... and a "dense" new-style group (libver latest, more than 8 links: fractal heap + v2 B-tree) */
    snprintf(path, sizeof path, "%s/refused_dense_group.h5", argv[1]);
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl); CHECK(f);
    for (int i = 0; i < 12; i++) { char nm[32]; hsize_t d1[1] = {2}; snprintf(nm, sizeof nm, "d%02d", i); dset_f32(f, nm, 1, d1, (unsigned)i, H5T_IEEE_F32LE, dcpl); }
    CHECK(H5Fclose(f));
    /* 7. a group wide enough for a two-level version-1 B-tree of symbol-table nodes (300 links: > 32 SNODs of <= 8 symbols) */
    snprintf(path, sizeof path, "%s/wide_group_two_level_btree.h5", argv[1]);
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT); CHECK(f);
    for (int i = 0; i < 300; i++) { char nm[32]; hsize_t d1[1] = {1}; snprintf(nm, sizeof nm, "w%03d:0", (i * 7) % 300); dset_f32(f, nm, 1, d1, (unsigned)((i * 7) % 300), H5T_IEEE_F32LE, dcpl); }
    CHECK(H5Fclose(f));
    H5Pclose(fapl); H5Pclose(dcpl);
    printf("wrote 7 fixtures under %s with libhdf5 %d.%d.%d\n", argv[1], H5_VERS_MAJOR, H5_VERS_MINOR, H5_VERS_RELEASE);
    return 0;
}
