package uni.bielefeld.cmg.reflexiv.gpu;

import scala.Tuple2;
import scala.Tuple4;

import java.io.Serializable;
import java.util.ArrayList;
import java.util.Iterator;
import java.util.List;

/**
 * A partition's reflexible k-mer records as flat arrays in the reference's own layout (rfx_records,
 * include/reflexiv_hip.h): key = the (k-1)-mer, marker 1 forward / 2 reflected, extension words
 * [extOff[i], extOff[i+1]) (word 0 under a sentinel bit, then 31 bases per word), left / right.
 * The two RDD element types of pipeline/ReflexivMain.java (:107-109) convert to and from it.
 */
public final class RfxRecords implements Serializable {
    public long n;
    public int keyWords = 1;
    public long[] key;
    public int[] marker;
    public long[] extOff;
    public long[] ext;
    public int[] left;
    public int[] right;

    /** room for capN records and capWords extension words */
    public RfxRecords(int capN, int capWords, int keyWords) {
        this.keyWords = Math.max(1, keyWords);
        final int a = Math.max(1, capN);
        key = new long[a * this.keyWords];
        marker = new int[a];
        extOff = new long[a + 1];
        ext = new long[Math.max(1, capWords)];
        left = new int[a];
        right = new int[a];
    }

    /** JavaPairRDD<Long, Tuple4<Integer, Long, Integer, Integer>> elements (single-word extensions) */
    public static RfxRecords fromSingle(Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> it) {
        List<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> l = new ArrayList<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>();
        while (it.hasNext()) l.add(it.next());
        RfxRecords r = new RfxRecords(l.size(), l.size(), 1);
        for (int i = 0; i < l.size(); i++) {
            Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>> t = l.get(i);
            r.key[i] = t._1; r.marker[i] = t._2._1(); r.ext[i] = t._2._2(); r.left[i] = t._2._3(); r.right[i] = t._2._4();
            r.extOff[i] = i;
        }
        r.extOff[l.size()] = l.size();
        r.n = l.size();
        return r;
    }

    /** JavaPairRDD<Long, Tuple4<Integer, Long[], Integer, Integer>> elements (array extensions) */
    public static RfxRecords fromArray(Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> it) {
        List<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> l = new ArrayList<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>>();
        int words = 0;
        while (it.hasNext()) { Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>> t = it.next(); words += t._2._2().length; l.add(t); }
        RfxRecords r = new RfxRecords(l.size(), words, 1);
        int w = 0;
        for (int i = 0; i < l.size(); i++) {
            Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>> t = l.get(i);
            r.key[i] = t._1; r.marker[i] = t._2._1(); r.left[i] = t._2._3(); r.right[i] = t._2._4();
            r.extOff[i] = w;
            for (Long x : t._2._2()) r.ext[w++] = x;
        }
        r.extOff[l.size()] = w;
        r.n = l.size();
        return r;
    }

    public Iterator<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> toSingle() {
        List<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>> l = new ArrayList<Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>>((int) n);
        for (int i = 0; i < n; i++)
            l.add(new Tuple2<Long, Tuple4<Integer, Long, Integer, Integer>>(key[i],
                    new Tuple4<Integer, Long, Integer, Integer>(marker[i], ext[(int) extOff[i]], left[i], right[i])));
        return l.iterator();
    }

    public Iterator<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> toArray() {
        List<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>> l = new ArrayList<Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>>((int) n);
        for (int i = 0; i < n; i++) {
            final int b = (int) extOff[i], e = (int) extOff[i + 1];
            Long[] x = new Long[e - b];
            for (int j = b; j < e; j++) x[j - b] = ext[j];
            l.add(new Tuple2<Long, Tuple4<Integer, Long[], Integer, Integer>>(key[i],
                    new Tuple4<Integer, Long[], Integer, Integer>(marker[i], x, left[i], right[i])));
        }
        return l.iterator();
    }

    public int words() { return (int) extOff[(int) n]; }
}
