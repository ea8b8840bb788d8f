package align2;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;

/**
 * The k-mer index on the device: AbstractIndex.findAdvanced (current/align2/AbstractIndex.java:83; sole call site
 * AbstractMapThread.java:736) batched over a read list.  build() runs IndexMaker4 + BBIndex.analyzeIndex + the genome-size tuning of
 * BBMap.loadIndex on the GPU from the chromosome arrays; the index then stays resident in HBM.
 *
 * The float-valued inputs of a read (makeKeyProbs, makeOffsets3, makeKeyScores, makeByteScoreArray, AbstractMapThread.java:659-728) stay
 * in Java: the caller passes offsets[], keyScores[] and baseScores[] exactly as quickMap computes them; the seam is integer-only.
 */
public final class BBIndexHIP implements AutoCloseable {

	static { System.loadLibrary("bbmap_amd_jni"); }

	public static final int PROFILE_BBMAP = 0, PROFILE_PACBIO = 1;
	private static final int READ_BYTES = 24, SITE_BYTES = 100, MAX_GAPS = 16;

	private static native long build(int device, int profile, int k, int chromBits, byte[][] chromArrays);
	private static native void destroy(long ctx);
	private static native void setMaxReadLen(long ctx, int maxLen);
	private static native void findBatch(long ctx, int nReads, ByteBuffer reads, ByteBuffer bases, ByteBuffer baseScores, int basesBytes,
			ByteBuffer keyinfo, int keyinfoInts, ByteBuffer sites, int maxSites, ByteBuffer nsites);

	long ctx;
	public final int profile;

	/** chromArrays[c] = Data.getChromosome(c).array for c = 1..numChroms (entry 0 unused); chromBits < 0 = automatic (BBMap.java:317-321) */
	public BBIndexHIP(int device, int profile_, int k, int chromBits, byte[][] chromArrays) {
		profile = profile_;
		ctx = build(device, profile, k, chromBits, chromArrays);
	}

	/** a sizing hint for the probe kernel (BBMap's maxReadLength), never a limit */
	public void setMaxReadLength(int len) { setMaxReadLen(ctx, len); }

	/** one site as the probe emits it (stream.SiteScore.java:999-1011) */
	public static final class Site {
		public int chrom, strand, start, stop, hits, score;
		public boolean perfect, semiperfect;
		public int[] gaps;
	}

	private int nReads, basesBytes, keyInts;
	private ByteBuffer reads = buf(200 * READ_BYTES), bases = buf(200 * 160), baseScores = buf(200 * 160), keyinfo = buf(200 * 64 * 4);
	private ByteBuffer sites, nsites;

	private static ByteBuffer buf(int bytes) { return ByteBuffer.allocateDirect(bytes).order(ByteOrder.LITTLE_ENDIAN); }
	private static ByteBuffer grown(ByteBuffer b, int used, int want) {
		if (b.capacity() >= want) { return b; }
		final ByteBuffer n = buf(Math.max(want, 2 * b.capacity()));
		b.position(0).limit(used); n.put(b); n.clear();
		return n;
	}

	public void clear() { nReads = 0; basesBytes = 0; keyInts = 0; }

	/** queue one read: basesP, baseScoresP (null without qualities is NOT accepted: pass makeByteScoreArray's output), offsets, keyScoresP */
	public int add(byte[] basesP, byte[] baseScoresP, int[] offsets, int[] keyScoresP) {
		reads = grown(reads, nReads * READ_BYTES, (nReads + 1) * READ_BYTES);
		bases = grown(bases, basesBytes, basesBytes + basesP.length);
		baseScores = grown(baseScores, basesBytes, basesBytes + basesP.length);
		keyinfo = grown(keyinfo, 4 * keyInts, 4 * (keyInts + 2 * offsets.length));
		final int o = nReads * READ_BYTES;
		reads.putLong(o, basesBytes); reads.putLong(o + 8, keyInts); reads.putInt(o + 16, basesP.length); reads.putInt(o + 20, offsets.length);
		bases.position(basesBytes); bases.put(basesP); bases.clear();
		baseScores.position(basesBytes); baseScores.put(baseScoresP, 0, basesP.length); baseScores.clear();
		for (int i = 0; i < offsets.length; i++) { keyinfo.putInt(4 * (keyInts + i), offsets[i]); }
		for (int i = 0; i < offsets.length; i++) { keyinfo.putInt(4 * (keyInts + offsets.length + i), keyScoresP[i]); }
		basesBytes += basesP.length; keyInts += 2 * offsets.length;
		return nReads++;
	}

	/** findAdvanced for every queued read; lists longer than maxSites come back as null (probe again with a larger maxSites) */
	public ArrayList<ArrayList<Site>> find(int maxSites) {
		final ArrayList<ArrayList<Site>> out = new ArrayList<ArrayList<Site>>(nReads);
		if (nReads == 0) { return out; }
		if (sites == null || sites.capacity() < nReads * maxSites * SITE_BYTES) { sites = buf(nReads * maxSites * SITE_BYTES); }
		if (nsites == null || nsites.capacity() < 4 * nReads) { nsites = buf(4 * nReads); }
		findBatch(ctx, nReads, reads, bases, baseScores, basesBytes, keyinfo, keyInts, sites, maxSites, nsites);
		for (int r = 0; r < nReads; r++) {
			final int n = nsites.getInt(4 * r);
			if (n < 0) { out.add(null); continue; }
			final ArrayList<Site> list = new ArrayList<Site>(n);
			for (int s = 0; s < n; s++) {
				final int o = (r * maxSites + s) * SITE_BYTES;
				final Site x = new Site();
				x.chrom = sites.getInt(o); x.strand = sites.getInt(o + 4); x.start = sites.getInt(o + 8); x.stop = sites.getInt(o + 12);
				x.hits = sites.getInt(o + 16); x.score = sites.getInt(o + 20);
				x.perfect = sites.getInt(o + 24) != 0; x.semiperfect = sites.getInt(o + 28) != 0;
				final int ng = Math.min(MAX_GAPS, sites.getInt(o + 32));
				if (ng > 0) { x.gaps = new int[ng]; for (int i = 0; i < ng; i++) { x.gaps[i] = sites.getInt(o + 36 + 4 * i); } }
				list.add(x);
			}
			out.add(list);
		}
		return out;
	}

	@Override
	public void close() { if (ctx != 0) { destroy(ctx); ctx = 0; } }
}
