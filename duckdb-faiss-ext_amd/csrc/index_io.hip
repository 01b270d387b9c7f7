// csrc/index_io.hip -- faiss::write_index / faiss::read_index (src/faiss_extension.cpp:199,234).
//
// FAISS's on-disk format [UPSTREAM: faiss/impl/index_write.cpp, index_read.cpp, impl/io_macros.h], restated:
//   little-endian; every index starts with a 4-byte fourcc (c0 | c1<<8 | c2<<16 | c3<<24) and the common header
//     int d; idx_t ntotal; idx_t dummy = 1<<20 (x2); bool is_trained (1 byte); int metric_type; [float metric_arg
//     only when metric_type > 1]
//   vectors are "size_t n" followed by n elements.
//   IxFI / IxF2 / IxFl  IndexFlatIP / IndexFlatL2 / IndexFlat : header, size_t nfloat, nfloat * f32
//   IxMp / IxM2         IndexIDMap / IndexIDMap2              : header, sub-index, vector<idx_t> id_map
//   IwFl                IndexIVFFlat : ivf header = {header, size_t nlist, size_t nprobe, quantizer index, direct map
//                       (char type, vector<idx_t> array [, hashtable pairs])}, then the inverted lists:
//                       "ilar", size_t nlist, size_t code_size, list sizes ("full": vector<size_t> of nlist sizes, or
//                       "sprs": vector<size_t> of (list_no, size) pairs), then per non-empty list codes, then ids
//   IHNf                IndexHNSWFlat : header, struct HNSW {vector<double> assign_probas, vector<int>
//                       cum_nneighbor_per_level, vector<int> levels, vector<size_t> offsets, vector<int32> neighbors,
//                       int32 entry_point, int max_level, int efConstruction, int efSearch, int upper_beam(=1)},
//                       then the storage index
// No .index file written by FAISS itself exists in the reference or in this image, so byte compatibility is
// "restated, unverified against a real file" (DESIGN.md); the round trip through this reader is tested.
#include "index.h"

#include <cerrno>
#include <cstring>

namespace mvs {

namespace {

constexpr uint32_t fourcc(const char (&s)[5]) {
	return (uint32_t)(unsigned char)s[0] | ((uint32_t)(unsigned char)s[1] << 8) | ((uint32_t)(unsigned char)s[2] << 16) |
	       ((uint32_t)(unsigned char)s[3] << 24);
}

struct Writer {
	FILE *f;
	const char *name;
	void raw(const void *p, size_t n) {
		if (n && fwrite(p, 1, n, f) != n)
			throw_faiss("void faiss::write_index(const faiss::Index*, const char*)", "faiss/impl/index_write.cpp",
			            "write error in %s: %s", name, strerror(errno));
	}
	template <typename T>
	void one(const T &v) {
		raw(&v, sizeof(T));
	}
	template <typename T>
	void vec(const std::vector<T> &v) {
		const uint64_t n = v.size();
		one(n);
		raw(v.data(), n * sizeof(T));
	}
};

struct Reader {
	FILE *f;
	const char *name;
	void raw(void *p, size_t n) {
		if (n && fread(p, 1, n, f) != n)
			throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
			            "read error in %s: %s", name, feof(f) ? "unexpected end of file" : strerror(errno));
	}
	template <typename T>
	void one(T &v) {
		raw(&v, sizeof(T));
	}
	template <typename T>
	void vec(std::vector<T> &v) {
		uint64_t n = 0;
		one(n);
		if (n > ((uint64_t)1 << 40) / sizeof(T))
			throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
			            "Error: 'size >= 0 && size < (uint64_t{1} << 40)' failed in %s", name);
		v.resize((size_t)n);
		raw(v.data(), (size_t)n * sizeof(T));
	}
};

void write_header(Writer &w, const HostIndex &h) {
	const int32_t d = h.d;
	const int64_t ntotal = h.ntotal, dummy = 1 << 20;
	const uint8_t trained = h.is_trained ? 1 : 0;
	const int32_t metric = h.metric;
	w.one(d);
	w.one(ntotal);
	w.one(dummy);
	w.one(dummy);
	w.one(trained);
	w.one(metric);
	if (metric > 1) {
		const float metric_arg = h.metric_arg;
		w.one(metric_arg);
	}
}
void read_header(Reader &r, HostIndex &h) {
	int32_t d = 0, metric = 0;
	int64_t ntotal = 0, dummy = 0;
	uint8_t trained = 0;
	r.one(d);
	r.one(ntotal);
	r.one(dummy);
	r.one(dummy);
	r.one(trained);
	r.one(metric);
	if (metric > 1) {
		float metric_arg;
		r.one(metric_arg);
		h.metric_arg = metric_arg;
	}
	h.d = d;
	h.ntotal = ntotal;
	h.is_trained = trained != 0;
	h.metric = metric;
}

void write_image(Writer &w, const HostIndex &h) {
	switch (h.kind) {
	case MVS_KIND_FLAT: {
		w.one(h.metric == METRIC_IP ? fourcc("IxFI") : (h.metric == METRIC_L2 ? fourcc("IxF2") : fourcc("IxFl")));
		write_header(w, h);
		w.vec(h.rows); // WRITEXBVECTOR: count of floats, then the codes
		return;
	}
	case MVS_KIND_IDMAP: {
		w.one(h.idmap2 ? fourcc("IxM2") : fourcc("IxMp"));
		write_header(w, h);
		write_image(w, *h.sub);
		w.vec(h.ids);
		return;
	}
	case MVS_KIND_IVFFLAT: {
		w.one(fourcc("IwFl"));
		write_header(w, h);
		const uint64_t nlist = (uint64_t)h.nlist, nprobe = (uint64_t)h.nprobe;
		w.one(nlist);
		w.one(nprobe);
		write_image(w, *h.sub);
		const char direct_map_type = 0; // DirectMap::NoMap
		w.one(direct_map_type);
		w.vec(std::vector<int64_t>());
		// write_InvertedLists (ArrayInvertedLists)
		w.one(fourcc("ilar"));
		w.one(nlist);
		const uint64_t code_size = (uint64_t)h.d * sizeof(float);
		w.one(code_size);
		uint64_t n_non0 = 0;
		for (const auto &l : h.list_ids)
			n_non0 += l.empty() ? 0 : 1;
		std::vector<uint64_t> sizes;
		if (n_non0 > nlist / 2) {
			w.one(fourcc("full"));
			for (const auto &l : h.list_ids)
				sizes.push_back(l.size());
		} else {
			w.one(fourcc("sprs"));
			for (size_t i = 0; i < h.list_ids.size(); i++)
				if (!h.list_ids[i].empty()) {
					sizes.push_back(i);
					sizes.push_back(h.list_ids[i].size());
				}
		}
		w.vec(sizes);
		for (size_t i = 0; i < h.list_ids.size(); i++)
			if (!h.list_ids[i].empty()) {
				w.raw(h.list_codes[i].data(), h.list_codes[i].size() * sizeof(float));
				w.raw(h.list_ids[i].data(), h.list_ids[i].size() * sizeof(int64_t));
			}
		return;
	}
	case MVS_KIND_HNSW: {
		w.one(fourcc("IHNf"));
		write_header(w, h);
		w.vec(h.assign_probas);
		w.vec(h.cum_nneighbor_per_level);
		w.vec(h.levels);
		w.vec(h.offsets);
		w.vec(h.neighbors);
		w.one(h.entry_point);
		const int32_t ml = h.max_level, efc = h.efConstruction, efs = h.efSearch, upper_beam = 1;
		w.one(ml);
		w.one(efc);
		w.one(efs);
		w.one(upper_beam);
		write_image(w, *h.sub);
		return;
	}
	}
	throw_faiss("void faiss::write_index(const faiss::Index*, const char*)", "faiss/impl/index_write.cpp",
	            "don't know how to serialize this type of index");
}

void read_image(Reader &r, HostIndex &h) {
	uint32_t cc = 0;
	r.one(cc);
	if (cc == fourcc("IxFI") || cc == fourcc("IxF2") || cc == fourcc("IxFl")) {
		h.kind = MVS_KIND_FLAT;
		read_header(r, h);
		r.vec(h.rows);
		if ((int64_t)h.rows.size() != h.ntotal * h.d)
			throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
			            "Error: 'idxf->codes.size() == idxf->ntotal * idxf->code_size' failed");
		return;
	}
	if (cc == fourcc("IxMp") || cc == fourcc("IxM2")) {
		h.kind = MVS_KIND_IDMAP;
		h.idmap2 = cc == fourcc("IxM2");
		read_header(r, h);
		h.sub.reset(new HostIndex);
		read_image(r, *h.sub);
		r.vec(h.ids);
		return;
	}
	if (cc == fourcc("IwFl")) {
		h.kind = MVS_KIND_IVFFLAT;
		read_header(r, h);
		uint64_t nlist = 0, nprobe = 0;
		r.one(nlist);
		r.one(nprobe);
		h.nlist = (int64_t)nlist;
		h.nprobe = (int64_t)nprobe;
		h.sub.reset(new HostIndex);
		read_image(r, *h.sub);
		char dm_type = 0;
		r.one(dm_type);
		std::vector<int64_t> dm_array;
		r.vec(dm_array);
		if (dm_type == 2) { // DirectMap::Hashtable: vector of (idx_t, idx_t) pairs
			std::vector<int64_t> pairs;
			uint64_t n = 0;
			r.one(n);
			if (n > ((uint64_t)1 << 40) / (2 * sizeof(int64_t))) // same bound READVECTOR applies
				throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
				            "Error: 'size >= 0 && size < (uint64_t{1} << 40)' failed in %s", r.name);
			pairs.resize((size_t)n * 2);
			r.raw(pairs.data(), pairs.size() * sizeof(int64_t));
		}
		uint32_t il = 0;
		r.one(il);
		h.list_ids.assign((size_t)nlist, {});
		h.list_codes.assign((size_t)nlist, {});
		if (il == fourcc("il00")) // no inverted lists stored
			return;
		if (il != fourcc("ilar"))
			throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
			            "read_InvertedLists: unsupported invlist type (only ArrayInvertedLists is implemented)");
		uint64_t nl2 = 0, code_size = 0;
		r.one(nl2);
		r.one(code_size);
		if (nl2 != nlist || code_size != (uint64_t)h.d * sizeof(float))
			throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
			            "inverted lists do not match the IVFFlat header");
		uint32_t list_type = 0;
		r.one(list_type);
		std::vector<uint64_t> sizes((size_t)nlist, 0), tmp;
		r.vec(tmp);
		if (list_type == fourcc("full")) {
			if (tmp.size() != nlist)
				throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
				            "Error: 'sizes.size() == nlist' failed");
			sizes = tmp;
		} else if (list_type == fourcc("sprs")) {
			for (size_t j = 0; j + 1 < tmp.size(); j += 2) {
				if (tmp[j] >= nlist)
					throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
					            "sparse list number out of range");
				sizes[(size_t)tmp[j]] = tmp[j + 1];
			}
		} else {
			throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
			            "list_type %ud not recognized", list_type);
		}
		for (size_t i = 0; i < (size_t)nlist; i++) {
			if (!sizes[i])
				continue;
			if (sizes[i] > ((uint64_t)1 << 40) / ((uint64_t)h.d * sizeof(float)))
				throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
				            "inverted list %zu: size %llu out of range in %s", i, (unsigned long long)sizes[i], r.name);
			h.list_codes[i].resize((size_t)sizes[i] * h.d);
			h.list_ids[i].resize((size_t)sizes[i]);
			r.raw(h.list_codes[i].data(), h.list_codes[i].size() * sizeof(float));
			r.raw(h.list_ids[i].data(), h.list_ids[i].size() * sizeof(int64_t));
		}
		return;
	}
	if (cc == fourcc("IHNf")) {
		h.kind = MVS_KIND_HNSW;
		read_header(r, h);
		r.vec(h.assign_probas);
		r.vec(h.cum_nneighbor_per_level);
		r.vec(h.levels);
		r.vec(h.offsets);
		r.vec(h.neighbors);
		r.one(h.entry_point);
		int32_t ml = 0, efc = 0, efs = 0, upper_beam = 0;
		r.one(ml);
		r.one(efc);
		r.one(efs);
		r.one(upper_beam);
		h.max_level = ml;
		h.efConstruction = efc;
		h.efSearch = efs;
		h.sub.reset(new HostIndex);
		read_image(r, *h.sub);
		return;
	}
	char txt[5] = {(char)(cc & 0xff), (char)((cc >> 8) & 0xff), (char)((cc >> 16) & 0xff), (char)((cc >> 24) & 0xff), 0};
	for (char &c : txt)
		if (c && (c < 32 || c > 126))
			c = '?';
	throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
	            "Index type 0x%08x (\"%s\") not recognized or not implemented on the MI355X path", cc, txt);
}

} // namespace

void write_index_file(IndexBase *ix, const char *filename) {
	HostIndex h;
	ix->to_host(h);
	FILE *f = fopen(filename, "wb");
	if (!f)
		throw_faiss("faiss::FileIOWriter::FileIOWriter(const char*)", "faiss/impl/io.cpp",
		            "could not open %s for writing: %s", filename, strerror(errno));
	Writer w {f, filename};
	try {
		write_image(w, h);
	} catch (...) {
		fclose(f);
		throw;
	}
	if (fclose(f) != 0)
		throw_faiss("faiss::FileIOWriter::~FileIOWriter()", "faiss/impl/io.cpp", "file %s close error: %s", filename,
		            strerror(errno));
}

IndexBase *read_index_file(const char *filename) {
	FILE *f = fopen(filename, "rb");
	if (!f)
		throw_faiss("faiss::FileIOReader::FileIOReader(const char*)", "faiss/impl/io.cpp",
		            "could not open %s for reading: %s", filename, strerror(errno));
	HostIndex h;
	Reader r {f, filename};
	try {
		read_image(r, h);
	} catch (...) {
		fclose(f);
		throw;
	}
	fclose(f);
	// env MVS_DEVICES=0,1,...: the loaded index is spread over those devices (csrc/sharded.hip)
	const std::vector<int> devs = shard_devices_from_env();
	if (devs.size() > 1)
		return shard_from_host(h, devs);
	return index_from_host(h, -1);
}

} // namespace mvs
