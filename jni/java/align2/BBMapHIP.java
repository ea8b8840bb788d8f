package align2;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

/**
 * The whole per-batch flow on the device: everything between AbstractIndex.findAdvanced and the end of
 * BBMapThread.processRead / processReadPair (current/align2/BBMapThread.java:389-732, :943-1362) -- probe, pairing, trimList,
 * scoreNoIndels, tip deletions, the scoreSlow rounds, rescue, and the final alignment stage (final pairing, ambiguity policy,
 * genMatchString -> realign_new, clipping, penalties) -- for one read list per call.  What comes back per read is its final record
 * ({@link #finalRecord}: mapped / chrom / strand / start / stop / mapScore / paired / ambiguous / perfect / rescued and the match
 * string, i.e. what Read.setFromSite + genMatchString leave in the Read) and its SiteScore list (start/stop, score/quickScore/
 * slowScore/pairedScore, hits, perfect/semiperfect, rescued, gaps); the Java side continues with statistics and output.
 *
 * paired: reads 2p and 2p+1 are mates.  Defaults are bbmap.sh's or mapPacBio.sh's (BBMap.setDefaults / BBMapPacBio.setDefaults).
 */
public final class BBMapHIP implements AutoCloseable {

	static { System.loadLibrary("bbmap_amd_jni"); }

	private static final int READ_BYTES = 24, MSITE_BYTES = 128, MAX_GAPS = 16, FINAL_BYTES = 64;
	/** nsites values below zero */
	public static final int NSITES_OVERFLOW = -1, NSITES_MATE_OVERFLOW = -2;

	private static native long create(long index, int profile, boolean paired, int maxReads, int maxReadLen, int maxSites);
	private static native void destroy(long ctx);
	private static native long mapBatch(long ctx, int nReads, ByteBuffer reads, ByteBuffer bases, ByteBuffer baseScores, int basesBytes,
			ByteBuffer keyinfo, int keyinfoInts, ByteBuffer nsites, ByteBuffer offsets, ByteBuffer sites, int sitesCap);
	private static native long getFinal(long ctx, int nReads, ByteBuffer records, ByteBuffer match, int matchCap);
	private static native int lastError(byte[] buf);

	private long ctx;
	private final BBIndexHIP index;      // borrowed: must outlive this object
	public final boolean paired;
	public final int maxReads;

	public BBMapHIP(BBIndexHIP index_, boolean paired_, int maxReads_, int maxReadLen, int maxSites) {
		index = index_; paired = paired_; maxReads = maxReads_;
		ctx = create(index.ctx, index.profile, paired, maxReads, maxReadLen, maxSites);
		nsites = buf(4 * maxReads); offsets = buf(8 * (maxReads + 1));
		sitesCap = 8 * maxReads; sites = buf(sitesCap * MSITE_BYTES);
	}

	/** stream.Read's mapping fields after processRead / processReadPair (what the SAM writer prints) */
	public static final class Final {
		public boolean mapped, paired, ambiguous, perfect, rescued;
		public int chrom, strand, start, stop, mapScore;
		/** Read.match in long format (m S N D I X Y C), null when the read is not mapped */
		public byte[] match;
	}

	/** stream.SiteScore, field for field */
	public static final class Site {
		public int chrom, strand, start, stop, hits, quickScore, score, slowScore, pairedScore;
		public boolean perfect, semiperfect, rescued;
		public int[] gaps;
	}

	private int nReads, basesBytes, keyInts, sitesCap;
	private ByteBuffer reads = buf(400 * READ_BYTES), bases = buf(400 * 160), baseScores = buf(400 * 160), keyinfo = buf(400 * 64 * 4);
	private ByteBuffer nsites, offsets, sites;
	private ByteBuffer finals = buf(400 * FINAL_BYTES), matches = buf(400 * 192);
	private boolean haveFinals;
	private long total;

	private static ByteBuffer buf(int bytes) { return ByteBuffer.allocateDirect(bytes).order(ByteOrder.LITTLE_ENDIAN); }
	private static ByteBuffer grown(ByteBuffer b, int used, int want) {
		if (b.capacity() >= want) { return b; }
		final ByteBuffer n = buf(Math.max(want, 2 * b.capacity()));
		b.position(0).limit(used); n.put(b); n.clear();
		return n;
	}

	public void clear() { nReads = 0; basesBytes = 0; keyInts = 0; haveFinals = false; }

	/** queue one read (a pair: r1 then r2); arguments as quickMap computes them (AbstractMapThread.java:659-728) */
	public int add(byte[] basesP, byte[] baseScoresP, int[] offsets_, int[] keyScoresP) {
		reads = grown(reads, nReads * READ_BYTES, (nReads + 1) * READ_BYTES);
		bases = grown(bases, basesBytes, basesBytes + basesP.length);
		baseScores = grown(baseScores, basesBytes, basesBytes + basesP.length);
		keyinfo = grown(keyinfo, 4 * keyInts, 4 * (keyInts + 2 * offsets_.length));
		final int o = nReads * READ_BYTES;
		reads.putLong(o, basesBytes); reads.putLong(o + 8, keyInts); reads.putInt(o + 16, basesP.length); reads.putInt(o + 20, offsets_.length);
		bases.position(basesBytes); bases.put(basesP); bases.clear();
		baseScores.position(basesBytes); baseScores.put(baseScoresP, 0, basesP.length); baseScores.clear();
		for (int i = 0; i < offsets_.length; i++) { keyinfo.putInt(4 * (keyInts + i), offsets_[i]); }
		for (int i = 0; i < offsets_.length; i++) { keyinfo.putInt(4 * (keyInts + offsets_.length + i), keyScoresP[i]); }
		basesBytes += basesP.length; keyInts += 2 * offsets_.length;
		return nReads++;
	}

	/** maps everything queued since clear() */
	public void map() {
		if (nReads == 0) { total = 0; return; }
		total = mapBatch(ctx, nReads, reads, bases, baseScores, basesBytes, keyinfo, keyInts, nsites, offsets, sites, sitesCap);
		if (total > sitesCap) {                   // the packed lists did not fit: once more with room for all of them
			sitesCap = (int)Math.min(Integer.MAX_VALUE / MSITE_BYTES, total + total / 4);
			sites = buf(sitesCap * MSITE_BYTES);
			total = mapBatch(ctx, nReads, reads, bases, baseScores, basesBytes, keyinfo, keyInts, nsites, offsets, sites, sitesCap);
		}
	}

	/** the final record of read r after map() (fetched from the device on first use) */
	public Final finalRecord(int r) {
		if (!haveFinals) {
			finals = grown(finals, 0, nReads * FINAL_BYTES);
			long need = getFinal(ctx, nReads, finals, matches, matches.capacity());
			if (need > matches.capacity()) {
				matches = buf((int)Math.min(Integer.MAX_VALUE - 64, need + need / 8));
				getFinal(ctx, nReads, finals, matches, matches.capacity());
			}
			haveFinals = true;
		}
		final int o = r * FINAL_BYTES;
		final Final f = new Final();
		f.mapped = finals.getInt(o) != 0; f.chrom = finals.getInt(o + 4); f.strand = finals.getInt(o + 8); f.start = finals.getInt(o + 12);
		f.stop = finals.getInt(o + 16); f.mapScore = finals.getInt(o + 20); f.paired = finals.getInt(o + 24) != 0;
		f.ambiguous = finals.getInt(o + 28) != 0; f.perfect = finals.getInt(o + 32) != 0; f.rescued = finals.getInt(o + 36) != 0;
		final int len = finals.getInt(o + 40);
		if (len > 0) {
			f.match = new byte[len];
			final ByteBuffer m = matches.duplicate();
			m.position((int)finals.getLong(o + 48));
			m.get(f.match, 0, len);
		}
		return f;
	}

	/** sites of read r after map(): >= 0, or NSITES_OVERFLOW / NSITES_MATE_OVERFLOW (the list fitted no tier: the read is unmapped) */
	public int numSites(int r) { return nsites.getInt(4 * r); }

	public Site site(int r, int s) {
		final int o = (int)(offsets.getLong(8 * r) + s) * MSITE_BYTES;
		final Site x = new Site();
		x.chrom = sites.getInt(o); x.strand = sites.getInt(o + 4); x.start = sites.getInt(o + 8); x.stop = sites.getInt(o + 12);
		x.hits = sites.getInt(o + 16); x.quickScore = sites.getInt(o + 20); x.score = sites.getInt(o + 24); x.slowScore = sites.getInt(o + 28);
		x.pairedScore = sites.getInt(o + 32); x.perfect = sites.getInt(o + 36) != 0; x.semiperfect = sites.getInt(o + 40) != 0;
		x.rescued = sites.getInt(o + 44) != 0;
		final int ng = Math.min(MAX_GAPS, sites.getInt(o + 48));
		if (ng > 0) { x.gaps = new int[ng]; for (int i = 0; i < ng; i++) { x.gaps[i] = sites.getInt(o + 52 + 4 * i); } }
		return x;
	}

	public static String lastErrorText() {
		final byte[] b = new byte[512];
		final int n = lastError(b);
		return new String(b, 0, n, java.nio.charset.StandardCharsets.UTF_8);
	}

	@Override
	public void close() { if (ctx != 0) { destroy(ctx); ctx = 0; } }
}
